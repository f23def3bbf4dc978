set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_mmgan_gpu.py tests/test_trainer_parity_bf16_gpu.py tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r3_t11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t11.log
tail -12 gpurun_out/r3_t11.log
python bench.py --workload mmgan --no-cpu-baseline --no-secondary > gpurun_out/bench_mmgan.json 2> gpurun_out/bench_mmgan.err; cut -c1-230 gpurun_out/bench_mmgan.json
python bench.py --workload mmgan --batch 16 --no-cpu-baseline --no-secondary > gpurun_out/bench_mmgan_b16.json 2>> gpurun_out/bench_mmgan.err; cut -c1-230 gpurun_out/bench_mmgan_b16.json
