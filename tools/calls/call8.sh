mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t8.log; tail -4 gpurun_out/r2_t8.log
python bench.py --no-cpu-baseline > gpurun_out/r2_b8_graph.json 2>gpurun_out/r2_b8.err
# the self-launcher end to end on the GPU box: 2 ranks on the one device over gloo (RCCL refuses two ranks on one GPU)
GDM_DIST_BACKEND=gloo GDM_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-roofline > gpurun_out/r2_b8_2rank_simnn.json 2>gpurun_out/r2_b8_2rank_simnn.err
GDM_DIST_BACKEND=gloo GDM_SINGLE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-roofline --workload mmgan > gpurun_out/r2_b8_2rank_mmgan.json 2>gpurun_out/r2_b8_2rank_mmgan.err
grep -h metric gpurun_out/r2_b8_*.json | cut -c1-900
tail -3 gpurun_out/r2_b8_2rank_simnn.err gpurun_out/r2_b8_2rank_mmgan.err | cut -c1-300
