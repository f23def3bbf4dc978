mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t9.log; tail -6 gpurun_out/r2_t9.log
python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2_b9_graph.json 2>gpurun_out/r2_b9.err
python bench.py --no-cpu-baseline --no-roofline --no-graph > gpurun_out/r2_b9_eager.json 2>>gpurun_out/r2_b9.err
python bench.py --no-cpu-baseline --no-roofline --mode elided > gpurun_out/r2_b9_elided.json 2>>gpurun_out/r2_b9.err
grep -h metric gpurun_out/r2_b9_*.json | cut -c1-200
