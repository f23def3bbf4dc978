set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_trunk" > gpurun_out/r3_t5.log 2>&1; rc=$?; tail -15 gpurun_out/r3_t5.log
if [ $rc -ne 0 ]; then exit 1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_simnn_eager -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph --no-overlap > gpurun_out/final_simnn_eager.log 2>&1
python tools/trace_split.py gpurun_out/final_simnn_eager 6
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
cut -c1-260 gpurun_out/bench_default.json
