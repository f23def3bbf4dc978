# one gpurun call: GPU suite, knock-out sweep of the fused conv2 backward, phase stamps, bench variants
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t3.log; tail -5 gpurun_out/r2_t3.log
tools/ko_sweep.sh run > gpurun_out/r2_ko1.log 2>&1
for w in bwd bww fwd; do GDM_LIB_TAG=stamps python tools/stamps.py $w >> gpurun_out/r2_stamps1.log 2>&1; done
python bench.py --no-cpu-baseline > gpurun_out/r2_b_graph.json 2>&1
python bench.py --no-cpu-baseline --no-graph > gpurun_out/r2_b_eager.json 2>&1
python bench.py --no-cpu-baseline --dtype fp32 > gpurun_out/r2_b_fp32.json 2>&1
python bench.py --no-cpu-baseline --workload mmgan > gpurun_out/r2_b_mm_graph.json 2>&1
python bench.py --no-cpu-baseline --workload mmgan --no-graph > gpurun_out/r2_b_mm_eager.json 2>&1
tail -n 3 gpurun_out/r2_b_*.json | cut -c1-300
cat gpurun_out/r2_ko1.log
