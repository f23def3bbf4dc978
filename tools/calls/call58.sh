cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do GDM_BENCH_STEP_TIMES=1 python bench.py --workload mmgan --no-graph --no-cpu-baseline --no-roofline 2>&1 | grep -v Warn | grep -E "host ms|metric" | cut -c1-330; done
