cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=256 python tools/bench_gemm.py 2>&1 | grep -v Warn | grep -E "fwd"
