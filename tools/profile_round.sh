# Round-end evidence in one call: kernel stats (graph + eager + model 2), HBM traffic passes, SQ counter passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_final.sh
bash tools/pmc_traffic.sh
python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/hbm_traffic.json > gpurun_out/hbm_traffic.txt
python tools/step_breakdown.py gpurun_out/final_simnn_eager > gpurun_out/step_breakdown.txt
bash tools/pmc_simnn.sh > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > gpurun_out/pmc_sq.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --workload mmgan > gpurun_out/bench_mmgan.json 2>> gpurun_out/bench_default.err
