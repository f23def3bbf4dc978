cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=512 NB=768 GDM_LIB_TAG=stamps timeout -k 10 300 python tools/stamps.py bww
B=512 NB=768 GDM_LIB_TAG=stamps timeout -k 10 300 python tools/stamps.py fwd
