cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fp32_eager -- python bench.py --dtype fp32 --steps 6 --warmup 2 --prime 0 --no-cpu-baseline --no-secondary --no-graph --no-overlap > gpurun_out/fp32_eager.log 2>&1
python tools/trace_split.py gpurun_out/fp32_eager 14
