cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "mmgan or linear or trainer_parity or dp" 2>&1 | tail -3
python tools/bench_linear_bn.py 2>&1 | grep -v Warn | tail -8
for i in 1 2; do python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
for i in 1 2; do python bench.py --workload mmgan --batch 16 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
