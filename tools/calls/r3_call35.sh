cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_mmgan16 -- python bench.py --workload mmgan --batch 16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --prime 0 > gpurun_out/final_mmgan16.log 2>&1
python tools/graph_timeline.py gpurun_out/final_mmgan16
