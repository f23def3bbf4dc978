cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/p21
python -m pytest tests -m gpu -q -x -k "fused_generator" > gpurun_out/r2_t21.log 2>&1; tail -3 gpurun_out/r2_t21.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p21 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-overlap > gpurun_out/p21.log 2>&1
python tools/step_breakdown.py gpurun_out/p21 | head -36
find gpurun_out/p21 -name "*.db" -delete
