# full gpu suite + kernel stats + default bench after the MFMA conv1-dW epilogue
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t3.log
tail -4 gpurun_out/r3_t3.log
bash tools/profile_final.sh
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
cut -c1-260 gpurun_out/bench_default.json
