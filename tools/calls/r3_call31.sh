cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GDM_LIB_TAG=stamps B=256 NB=1024 timeout -k 10 120 python tools/stamps.py c1
