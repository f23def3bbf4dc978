mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r2_t18.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t18.log; tail -25 gpurun_out/r2_t18.log | cut -c1-220
