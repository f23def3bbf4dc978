cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "simnn or ops" > gpurun_out/r2_t41.log 2>&1; tail -3 gpurun_out/r2_t41.log
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/hbm_traffic.json | head -30
find gpurun_out -name "*.db" -delete
