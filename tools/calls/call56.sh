cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "simnn or mmgan or dp or trainer_parity" 2>&1 | tail -2
for i in 1 2 3 4 5; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c120-170; done
for i in 1 2 3; do python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c120-170; done
for i in 1 2 3; do python bench.py --workload mmgan --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | cut -c120-170; done
python bench.py --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | cut -c120-170
