# round-3 evidence in one call: gpu suite, kernel stats (graph / eager / model 2), HBM traffic, SQ counters, bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_r03.jsonl
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_r03.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r03.log
tail -3 gpurun_out/pytest_gpu_r03.log
bash tools/profile_final.sh
bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/hbm_traffic.json > gpurun_out/hbm_traffic.txt
python tools/step_breakdown.py gpurun_out/final_simnn_eager > gpurun_out/step_breakdown.txt
bash tools/pmc_simnn.sh > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_a gpurun_out/pmc_b > gpurun_out/pmc_sq.txt
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/bench_default_20.json 2>> gpurun_out/bench_default.err
python bench.py --workload mmgan > gpurun_out/bench_mmgan.json 2>> gpurun_out/bench_default.err
python bench.py --workload mmgan --batch 16 --no-cpu-baseline --no-secondary > gpurun_out/bench_mmgan_b16.json 2>> gpurun_out/bench_default.err
python bench.py --mode elided --no-cpu-baseline --no-secondary > gpurun_out/bench_simnn_elided.json 2>> gpurun_out/bench_default.err
python bench.py --no-pipeline --no-cpu-baseline --no-secondary > gpurun_out/bench_simnn_nopipeline.json 2>> gpurun_out/bench_default.err
python bench.py --no-graph --no-cpu-baseline --no-secondary > gpurun_out/bench_simnn_eager.json 2>> gpurun_out/bench_default.err
python bench.py --dtype fp32 --no-cpu-baseline --no-secondary > gpurun_out/bench_simnn_fp32.json 2>> gpurun_out/bench_default.err
python bench.py --dtype fp32 --no-pipeline --no-cpu-baseline --no-secondary > gpurun_out/bench_simnn_fp32_nopipeline.json 2>> gpurun_out/bench_default.err
python bench.py --batch 16 --width 64 --no-cpu-baseline --no-secondary > gpurun_out/bench_simnn_c1_b16_w64.json 2>> gpurun_out/bench_default.err
python bench.py --batch 16 --width 216 --no-cpu-baseline --no-secondary > gpurun_out/bench_simnn_c1_b16_w216.json 2>> gpurun_out/bench_default.err
python bench.py --batch 128 --width 216 --no-cpu-baseline --no-secondary > gpurun_out/bench_simnn_c5_b128_w216.json 2>> gpurun_out/bench_default.err
for f in bench_default bench_default_20 bench_mmgan bench_mmgan_b16 bench_simnn_elided bench_simnn_nopipeline bench_simnn_eager bench_simnn_fp32 bench_simnn_fp32_nopipeline bench_simnn_c1_b16_w64 bench_simnn_c1_b16_w216 bench_simnn_c5_b128_w216; do echo $f $(python -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print(d['ms_per_step'], d['value'], (d.get('roofline') or {}).get('frac'))"); done
