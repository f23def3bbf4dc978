set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_trunk" > gpurun_out/r3_t10.log 2>&1; rc=$?; tail -3 gpurun_out/r3_t10.log
if [ $rc -ne 0 ]; then exit 1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_simnn_eager -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph --no-overlap > gpurun_out/final_simnn_eager.log 2>&1
python tools/trace_split.py gpurun_out/final_simnn_eager 4
bash tools/pmc_simnn.sh > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > gpurun_out/pmc_sq.txt
grep -A4 "conv2_bwd" gpurun_out/pmc_sq.txt
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-230 gpurun_out/bench_default.json
