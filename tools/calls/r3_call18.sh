set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GDM_LIB_TAG=stamps timeout -k 10 300 python tools/experiments/adam_stamps.py
bash tools/calls/r3_call17.sh
