cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do echo "SLAB1=$v"; GDM_EXP_SLAB1=$v python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
done
GDM_EXP_SLAB1=1 python -m pytest tests -m gpu -q -x -k "simnn" 2>&1 | tail -2
