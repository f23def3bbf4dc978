cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 5000 --warmup 20 --no-secondary --no-cpu-baseline --no-roofline | cut -c1-260
python bench.py --workload mmgan --steps 5000 --warmup 20 --no-secondary --no-cpu-baseline --no-roofline | cut -c1-260
