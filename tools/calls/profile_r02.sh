# Round-2 evidence in one call: kernel stats (graph / single-stream eager / model 2), HBM traffic passes, SQ counters,
# bench lines (default, eager, elided, fp32, model 2 graph + eager).  Summaries are copied to profiles/ afterwards
# (tools/collect_profiles.py r02).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r02.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r02.log; tail -3 gpurun_out/pytest_gpu_r02.log
rm -rf gpurun_out/final_simnn gpurun_out/final_simnn_eager gpurun_out/final_mmgan gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_a gpurun_out/pmc_b
bash tools/profile_final.sh
bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/hbm_traffic.json > gpurun_out/hbm_traffic.txt
python tools/step_breakdown.py gpurun_out/final_simnn_eager > gpurun_out/step_breakdown.txt
bash tools/pmc_simnn.sh > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > gpurun_out/pmc_sq.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --workload mmgan > gpurun_out/bench_mmgan.json 2>> gpurun_out/bench_default.err
python bench.py --no-cpu-baseline --no-roofline --no-graph > gpurun_out/bench_simnn_eager.json 2>> gpurun_out/bench_default.err
python bench.py --no-cpu-baseline --no-roofline --mode elided > gpurun_out/bench_simnn_elided.json 2>> gpurun_out/bench_default.err
python bench.py --no-cpu-baseline --no-roofline --no-pipeline > gpurun_out/bench_simnn_nopipeline.json 2>> gpurun_out/bench_default.err
python bench.py --no-cpu-baseline --dtype fp32 > gpurun_out/bench_simnn_fp32.json 2>> gpurun_out/bench_default.err
python bench.py --no-cpu-baseline --no-roofline --workload mmgan --no-graph --steps 200 --warmup 20 > gpurun_out/bench_mmgan_eager.json 2>> gpurun_out/bench_default.err
python bench.py --no-cpu-baseline --no-roofline --workload mmgan --batch 16 > gpurun_out/bench_mmgan_b16.json 2>> gpurun_out/bench_default.err
python bench.py --batch 16 --width 64 --no-cpu-baseline --no-roofline > gpurun_out/bench_simnn_c1_b16_w64.json 2>> gpurun_out/bench_default.err
python bench.py --batch 16 --width 216 --no-cpu-baseline --no-roofline > gpurun_out/bench_simnn_c1_b16_w216.json 2>> gpurun_out/bench_default.err
python bench.py --batch 128 --width 216 --no-cpu-baseline --no-roofline > gpurun_out/bench_simnn_c5_b128_w216.json 2>> gpurun_out/bench_default.err
rm -rf gpurun_out/tl_mmgan
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_mmgan -- python bench.py --workload mmgan --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/tl_mmgan.log 2>&1
python tools/graph_timeline.py gpurun_out/tl_mmgan > gpurun_out/mmgan_replay_timeline.txt
find gpurun_out -name "*.db" -delete
grep -h metric gpurun_out/bench_*.json | cut -c1-210
cat gpurun_out/step_breakdown.txt | head -40
