cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p30; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p30 -- python bench.py --workload mmgan --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/p30.log 2>&1
python tools/graph_timeline.py gpurun_out/p30 | tail -40
find gpurun_out/p30 -name "*.db" -delete
