cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --workload mmgan --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
python bench.py --workload mmgan --no-graph --steps 100 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t34.log 2>&1; tail -4 gpurun_out/r2_t34.log
