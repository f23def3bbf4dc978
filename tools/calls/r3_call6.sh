set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_r03.jsonl
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t6.log
tail -25 gpurun_out/r3_t6.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_simnn_eager -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph --no-overlap > gpurun_out/final_simnn_eager.log 2>&1
python tools/trace_split.py gpurun_out/final_simnn_eager 6
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
cut -c1-260 gpurun_out/bench_default.json
cat gpurun_out/parity_r03.jsonl | cut -c1-400
