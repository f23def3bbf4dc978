set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pipe in 1 0; do for cap in 256 384 512; do echo "PIPE=$pipe CAP=$cap"; GDM_BD_PIPE=$pipe GDM_BD_CAP=$cap ONLY=conv2_bwd_fused timeout -k 10 120 python tools/bench_op.py || exit 1; done; done
