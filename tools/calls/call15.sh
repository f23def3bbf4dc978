cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; L=gpurun_out/r2_rep.log; : > $L
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-170 >> $L; done
python bench.py --no-cpu-baseline --no-roofline --workload mmgan 2>/dev/null | cut -c1-170 >> $L
rm -rf gpurun_out/p15
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p15 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-overlap > gpurun_out/p15.log 2>&1
python tools/step_breakdown.py gpurun_out/p15 >> $L 2>&1
find gpurun_out/p15 -name "*.db" -delete
cat $L | head -40
