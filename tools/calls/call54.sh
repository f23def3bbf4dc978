cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GDM_EXP_C2F_DYN=1 python -m pytest tests -m gpu -q -x -k "simnn or ops" 2>&1 | tail -3
for rep in 1 2; do
for v in 0 1; do echo "C2F_DYN=$v"; GDM_EXP_C2F_DYN=$v python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; GDM_EXP_C2F_DYN=$v python tools/bench_op.py 2>&1 | grep "conv2_fwd"; done
done
