cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 512 256; do for ns in 1 2 3 4 6; do echo "B=$b GDM_BW_NSEG=$ns $(GDM_BW_NSEG=$ns B=$b timeout -k 10 200 python tools/bench_op.py | grep conv2_bwd_weight)"; done; done
