mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t14.log; tail -4 gpurun_out/r2_t14.log
python bench.py --no-cpu-baseline --dtype fp32 > gpurun_out/r2_b14_fp32.json 2>gpurun_out/r2_b14.err
python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2_b14_graph.json 2>>gpurun_out/r2_b14.err
grep -h metric gpurun_out/r2_b14_*.json | cut -c1-1300
