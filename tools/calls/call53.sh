cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "simnn or trainer_parity" 2>&1 | tail -3
rm -rf gpurun_out/p53
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p53 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-overlap > gpurun_out/p53.log 2>&1
python tools/step_breakdown.py gpurun_out/p53 | grep -E "gen_|convt|bn_fin|span"
find gpurun_out/p53 -name "*.db" -delete
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
