cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GDM_BENCH_STEP_TIMES=1 python bench.py --workload mmgan --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | grep "host ms\|metric" | cut -c1-400
GDM_BENCH_STEP_TIMES=1 python bench.py --workload mmgan --batch 16 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | grep "host ms\|metric" | cut -c1-400
