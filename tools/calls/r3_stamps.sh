cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nb in 512; do
B=512 NB=$nb GDM_LIB_TAG=stamps timeout -k 10 300 python tools/stamps.py bwd
B=256 NB=$nb GDM_LIB_TAG=stamps timeout -k 10 300 python tools/stamps.py bwd
done
