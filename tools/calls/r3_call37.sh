cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/capsweep -- python bench.py --steps 10 --warmup 3 --prime 0 --no-cpu-baseline --no-secondary --no-graph --no-overlap > gpurun_out/capsweep.log 2>&1; python tools/trace_split.py gpurun_out/capsweep 4 | grep "conv2_fwd\|bwd_weight"; }
run GDM_X=0
run GDM_C2F_CAP=512
run GDM_C2F_CAP=1024
run GDM_BW_CAP=512
run GDM_BW_CAP=1024
run GDM_X=0
