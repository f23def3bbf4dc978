set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 256 512 768; do B=$b timeout -k 10 200 python tools/bench_gemm.py | grep -v "dh" || exit 1; done
for b in 256 512 768; do B=$b ONLY=conv2_fwd timeout -k 10 120 python tools/bench_op.py || exit 1;  B=$b ONLY=conv1_fwd timeout -k 10 120 python tools/bench_op.py || exit 1; done
