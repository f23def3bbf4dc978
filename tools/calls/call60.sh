cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GDM_BENCH_STEP_TIMES=1 python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>&1 | grep -v Warn | grep -E "host ms|metric" | cut -c1-260
GDM_BENCH_STEP_TIMES=1 python bench.py --workload mmgan --batch 16 --no-cpu-baseline --no-roofline 2>&1 | grep -v Warn | grep -E "host ms|metric" | cut -c1-260
GDM_BENCH_STEP_TIMES=1 python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep -v Warn | grep -E "host ms|metric" | cut -c1-260
