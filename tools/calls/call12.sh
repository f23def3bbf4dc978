cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; L=gpurun_out/r2_nopipe.log; : > $L
python -m pytest tests -m gpu -q -x -k "simnn or ops" > gpurun_out/r2_t12.log 2>&1; tail -3 gpurun_out/r2_t12.log >> $L
python bench.py --no-cpu-baseline --no-roofline --no-pipeline >> $L 2>&1
python bench.py --no-cpu-baseline --no-roofline --no-pipeline --no-graph >> $L 2>&1
python bench.py --no-cpu-baseline --no-roofline >> $L 2>&1
rm -rf gpurun_out/p12
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p12 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-pipeline > gpurun_out/p12.log 2>&1
python - <<'PY' >> $L
import csv,glob
f=glob.glob('gpurun_out/p12/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
find gpurun_out/p12 -name "*.db" -delete
grep -v amdgpu.ids $L | cut -c1-230
