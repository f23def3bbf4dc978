cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "fused_dcnn_kernel" 2>&1 | tail -15
