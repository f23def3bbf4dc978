# GEMM variant experiment: per-kernel GEMM times of one eager training step for variant 0, 1 and the host heuristic.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 auto; do
  if [ "$v" = auto ]; then unset GDM_GEMM_VARIANT; else export GDM_GEMM_VARIANT=$v; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gv_$v -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-graph --no-overlap > gpurun_out/gv_$v.log 2>&1
  echo "== variant $v"; python tools/step_breakdown.py gpurun_out/gv_$v detail | grep -i "gemm\|step span"
done
