# round-3 profile set: kernel stats (graph / eager / model 2), HBM traffic passes, SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_final.sh
bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/hbm_traffic.json > gpurun_out/hbm_traffic.txt
python tools/step_breakdown.py gpurun_out/final_simnn_eager > gpurun_out/step_breakdown.txt
bash tools/pmc_simnn.sh > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > gpurun_out/pmc_sq.txt
head -30 gpurun_out/pmc_sq.txt
head -12 gpurun_out/hbm_traffic.txt
