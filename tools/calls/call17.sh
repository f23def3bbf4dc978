mkdir -p gpurun_out
python tools/bench_adam_overlap.py 2>/dev/null | tee gpurun_out/r2_adam_overlap.log
