# Round-end evidence: per-kernel times (rocprofv3 --kernel-trace --stats) of the default bench command and of the
# model-2 bench, single-stream eager (per-kernel durations are meaningful) and graph+overlap (what bench.py reports).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_simnn -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --prime 0 > gpurun_out/final_simnn.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_simnn_eager -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --prime 0 --no-graph --no-overlap > gpurun_out/final_simnn_eager.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_mmgan -- python bench.py --workload mmgan --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --prime 0 > gpurun_out/final_mmgan.log 2>&1
grep -h metric gpurun_out/final_simnn.log gpurun_out/final_simnn_eager.log gpurun_out/final_mmgan.log | cut -c1-200
