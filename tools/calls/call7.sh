mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t7.log; tail -12 gpurun_out/r2_t7.log
L=gpurun_out/r2_mm_caps.log; : > $L
for cap in 224 256 192 128; do
  echo "== GDM_DCNN_CAP=$cap" >> $L
  GDM_DCNN_CAP=$cap python bench.py --no-cpu-baseline --no-roofline --workload mmgan 2>/dev/null | cut -c1-170 >> $L
done
echo "== eager (224)" >> $L; python bench.py --no-cpu-baseline --no-roofline --workload mmgan --no-graph 2>/dev/null | cut -c1-170 >> $L
echo "== B=16" >> $L; python bench.py --no-cpu-baseline --no-roofline --workload mmgan --batch 16 2>/dev/null | cut -c1-170 >> $L
cat $L
