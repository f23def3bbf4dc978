cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in "" c1d6 c1d8; do echo "lib=$tag $(GDM_LIB_TAG=$tag B=256 timeout -k 10 200 python tools/bench_op.py | grep conv1_fwd)"; done
