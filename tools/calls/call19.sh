mkdir -p gpurun_out
python tools/overlap_probe.py 2>/dev/null | tee gpurun_out/r2_overlap_probe.log
