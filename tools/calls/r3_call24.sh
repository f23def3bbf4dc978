set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_simnn_gpu.py tests/test_ops_gpu.py tests/test_trainer_parity_bf16_gpu.py tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r3_t24.log 2>&1; rc=$?; tail -5 gpurun_out/r3_t24.log; [ $rc -eq 0 ] || exit $rc
python bench.py --no-cpu-baseline --no-secondary --no-roofline | cut -c1-200
python bench.py --no-cpu-baseline --no-secondary --no-roofline --no-graph | cut -c1-200
