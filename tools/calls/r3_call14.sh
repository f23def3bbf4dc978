set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r3_t14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t14.log
tail -25 gpurun_out/r3_t14.log
