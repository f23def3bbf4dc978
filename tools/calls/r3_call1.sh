# round 3, call 1: the whole gpu suite after the ADVICE fixes, then the default bench line (new keys: loss_parity, secondary)
set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t1.log
tail -5 gpurun_out/r3_t1.log
python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r3_bench1_20.json 2>> gpurun_out/r3_bench1.err
cut -c1-400 gpurun_out/r3_bench1.json; cut -c1-300 gpurun_out/r3_bench1_20.json
