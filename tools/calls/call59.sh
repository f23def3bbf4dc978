cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python bench.py --workload mmgan --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | cut -c120-170; done
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c120-170; done
python bench.py 2>/dev/null | cut -c1-260
