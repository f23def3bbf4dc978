cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "mmgan or dcnn or linear or dp" > gpurun_out/r2_t28.log 2>&1; tail -3 gpurun_out/r2_t28.log
python tools/bench_linear_bn.py 2>&1 | grep -v Warn | tail -8
for i in 1 2 3; do python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
python bench.py --workload mmgan --no-cpu-baseline --no-roofline --no-graph 2>/dev/null | cut -c90-170
python bench.py --workload mmgan --batch 16 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170
