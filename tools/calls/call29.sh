cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "mmgan or dp" > gpurun_out/r2_t29.log 2>&1; tail -3 gpurun_out/r2_t29.log
for cap in 224 256; do echo "cap $cap"; GDM_DCNN_CAP=$cap python tools/bench_dcnn.py 2>&1 | grep -v Warn | grep "xa only, grad" | tail -2
for i in 1 2 3; do GDM_DCNN_CAP=$cap python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done; done
