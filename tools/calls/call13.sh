cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/p13
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p13 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph --no-overlap --dtype fp32 > gpurun_out/p13.log 2>&1
python tools/step_breakdown.py gpurun_out/p13 > gpurun_out/p13_breakdown.txt 2>&1
cat gpurun_out/p13_breakdown.txt | head -30; grep -h metric gpurun_out/p13.log | cut -c1-200
find gpurun_out/p13 -name "*.db" -delete
