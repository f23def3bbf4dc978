cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do for tag in old base; do
GDM_BENCH_ALLOW_EXPERIMENT=1 GDM_LIB_TAG=$tag rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$tag$i -- python bench.py --steps 10 --warmup 3 --prime 0 --no-cpu-baseline --no-secondary --no-graph --no-overlap > gpurun_out/ab.log 2>&1
echo "TAG=$tag"; python tools/trace_split.py gpurun_out/ab_$tag$i 4 | grep "conv1_fwd\|conv2_fwd"
done; done
