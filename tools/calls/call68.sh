cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c100-175
python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c100-175
