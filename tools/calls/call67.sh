cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p67
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p67 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-overlap > gpurun_out/p67.log 2>&1
python tools/step_breakdown.py gpurun_out/p67 | grep -E "gen_|convt|bn_fin|span"
find gpurun_out/p67 -name "*.db" -delete
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c100-175
GDM_EXP_GEN_GRAPH=0 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c100-175
