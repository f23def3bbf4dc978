cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=512 timeout -k 10 300 python tools/bench_gemm.py
B=256 timeout -k 10 300 python tools/bench_gemm.py
