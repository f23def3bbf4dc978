cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; L=gpurun_out/r2_rep2.log; : > $L
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t16.log; tail -3 gpurun_out/r2_t16.log >> $L
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-170 >> $L; done
rm -rf gpurun_out/p16
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p16 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-overlap > gpurun_out/p16.log 2>&1
python tools/step_breakdown.py gpurun_out/p16 >> $L 2>&1
find gpurun_out/p16 -name "*.db" -delete
cat $L | head -40
