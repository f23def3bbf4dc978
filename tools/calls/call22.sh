mkdir -p gpurun_out
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-170; done
python bench.py --no-cpu-baseline --no-roofline --no-graph 2>/dev/null | cut -c1-170
python tools/overlap_probe.py 2>/dev/null | head -5
