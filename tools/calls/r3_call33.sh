cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in "" d8 d2; do for cap in 1024 1536 2048 3072; do for b in 256 512; do echo -n "TAG=$tag CAP=$cap "; GDM_LIB_TAG=$tag GDM_C1_CAP=$cap B=$b ONLY=conv1_fwd timeout -k 10 120 python tools/bench_op.py | grep median || exit 1; done; done; done
