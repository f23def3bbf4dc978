cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p64
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p64 -- python bench.py --dtype fp32 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph --no-overlap > gpurun_out/p64.log 2>&1
python tools/step_breakdown.py gpurun_out/p64 | head -24
find gpurun_out/p64 -name "*.db" -delete
