cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cap in 256 384 512; do echo "GDM_BD_CAP=$cap"; GDM_BD_CAP=$cap B=512 timeout -k 10 200 python tools/bench_op.py | grep "conv2_bwd_fused"; done
