cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "simnn or trainer_parity or dp" > gpurun_out/r2_t37.log 2>&1; tail -4 gpurun_out/r2_t37.log
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
