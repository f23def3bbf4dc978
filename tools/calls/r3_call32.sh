set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_simnn_gpu.py tests/test_ops_gpu.py -m gpu -x -q > gpurun_out/r3_t32.log 2>&1; rc=$?; tail -3 gpurun_out/r3_t32.log; [ $rc -eq 0 ] || exit $rc
for b in 256 512; do B=$b ONLY=conv1_fwd timeout -k 10 120 python tools/bench_op.py; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_simnn_eager -- python bench.py --steps 10 --warmup 3 --prime 0 --no-cpu-baseline --no-secondary --no-graph --no-overlap > gpurun_out/final_simnn_eager.log 2>&1
python tools/trace_split.py gpurun_out/final_simnn_eager 4
python bench.py --no-cpu-baseline --no-secondary --no-roofline | cut -c1-200
