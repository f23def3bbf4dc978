set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_trunk" > gpurun_out/r3_t4.log 2>&1; rc=$?; tail -15 gpurun_out/r3_t4.log
if [ $rc -ne 0 ]; then exit 1; fi
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t4b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t4b.log
tail -4 gpurun_out/r3_t4b.log
bash tools/profile_final.sh
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
cut -c1-260 gpurun_out/bench_default.json
