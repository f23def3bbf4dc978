cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/p23
python -m pytest tests -m gpu -q -x -k "fused_generator or gen" > gpurun_out/r2_t23.log 2>&1; tail -3 gpurun_out/r2_t23.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p23 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-overlap > gpurun_out/p23.log 2>&1
python tools/step_breakdown.py gpurun_out/p23 | grep -E "gen_|convt|bn_fin|total" 
find gpurun_out/p23 -name "*.db" -delete
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-170; done
python tools/overlap_probe.py 2>/dev/null | head -5
