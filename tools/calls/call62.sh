cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null | cut -c1-900
python bench.py --workload mmgan --no-cpu-baseline 2>/dev/null | cut -c1-700
python bench.py --workload mmgan --batch 16 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170
