mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t20.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t20.log; tail -25 gpurun_out/r2_t20.log | cut -c1-250
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-170; done
python tools/overlap_probe.py 2>/dev/null | head -5
