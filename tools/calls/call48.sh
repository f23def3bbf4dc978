cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "mmgan or dcnn or trainer_parity" 2>&1 | tail -3
python tools/bench_dcnn.py 2>&1 | grep -v Warn | grep "xa only, grad"
for i in 1 2 3; do python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
