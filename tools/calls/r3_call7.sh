set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py tests/test_simnn_gpu.py -m gpu -x -q > gpurun_out/r3_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t7.log
tail -25 gpurun_out/r3_t7.log
python bench.py --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r3_b7_one.json 2> gpurun_out/r3_b7.err; cut -c1-200 gpurun_out/r3_b7_one.json
python bench.py --no-cpu-baseline --no-secondary --no-roofline --pieces > gpurun_out/r3_b7_pieces.json 2>> gpurun_out/r3_b7.err; cut -c1-200 gpurun_out/r3_b7_pieces.json
python bench.py --no-cpu-baseline --no-secondary --no-roofline --no-graph > gpurun_out/r3_b7_eager.json 2>> gpurun_out/r3_b7.err; cut -c1-200 gpurun_out/r3_b7_eager.json
GDM_DIST_BACKEND=gloo GDM_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r3_b7_2ranks.json 2>> gpurun_out/r3_b7.err; cut -c1-700 gpurun_out/r3_b7_2ranks.json
tail -5 gpurun_out/r3_b7.err
