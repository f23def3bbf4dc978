cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "mmgan or dcnn" > gpurun_out/r2_t26.log 2>&1; tail -3 gpurun_out/r2_t26.log
bash tools/calls/dcnn_stamps.sh
python tools/bench_dcnn.py 2>&1 | grep -v Warn | grep "xa only"
for i in 1 2; do python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-170; done
