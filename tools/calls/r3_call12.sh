set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t12.log
tail -12 gpurun_out/r3_t12.log
