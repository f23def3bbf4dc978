cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "mmgan or simnn or trainer_parity or dp" > gpurun_out/r2_t40.log 2>&1; tail -3 gpurun_out/r2_t40.log
for i in 1 2 3; do python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
python bench.py --workload mmgan --batch 16 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
