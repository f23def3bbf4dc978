set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -m gpu -x -q > gpurun_out/r3_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t8.log
tail -5 gpurun_out/r3_t8.log
python bench.py --no-cpu-baseline --no-secondary --no-roofline --pieces > gpurun_out/r3_b7_pieces.json 2>> gpurun_out/r3_b7.err; cut -c1-200 gpurun_out/r3_b7_pieces.json
python bench.py --no-cpu-baseline --no-secondary --no-roofline --no-graph > gpurun_out/r3_b7_eager.json 2>> gpurun_out/r3_b7.err; cut -c1-200 gpurun_out/r3_b7_eager.json
