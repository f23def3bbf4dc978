cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "mmgan or dp or trainer_parity or abi" > gpurun_out/r2_t32.log 2>&1; tail -5 gpurun_out/r2_t32.log
for i in 1 2 3; do python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
python bench.py --workload mmgan --batch 16 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170
python bench.py --workload mmgan --no-graph --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170
rm -rf gpurun_out/p30
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p30 -- python bench.py --workload mmgan --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/p30.log 2>&1
python tools/graph_timeline.py gpurun_out/p30 | tail -16
find gpurun_out/p30 -name "*.db" -delete
