mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r2_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t5.log; tail -15 gpurun_out/r2_t5.log
python bench.py --no-cpu-baseline > gpurun_out/r2_b5_graph.json 2>&1
python bench.py --no-cpu-baseline --no-roofline --no-graph > gpurun_out/r2_b5_eager.json 2>&1
python bench.py --no-cpu-baseline --no-roofline --dtype fp32 > gpurun_out/r2_b5_fp32.json 2>&1
python bench.py --no-cpu-baseline --no-roofline --workload mmgan > gpurun_out/r2_b5_mm.json 2>&1
grep -h metric gpurun_out/r2_b5_*.json | cut -c1-200
tail -3 gpurun_out/r2_b5_fp32.json | cut -c1-300
