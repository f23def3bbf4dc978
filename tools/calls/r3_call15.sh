set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GDM_DIST_BACKEND=gloo GDM_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --prime 3 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r3_b15_simnn2.json 2> gpurun_out/r3_b15.err; cut -c1-600 gpurun_out/r3_b15_simnn2.json
GDM_DIST_BACKEND=gloo GDM_SINGLE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --workload mmgan --steps 10 --warmup 3 --prime 3 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r3_b15_mmgan2.json 2>> gpurun_out/r3_b15.err; cut -c1-600 gpurun_out/r3_b15_mmgan2.json
tail -3 gpurun_out/r3_b15.err
