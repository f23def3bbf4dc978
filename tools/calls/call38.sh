cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do echo "NO_DW_FORK=$v"; GDM_EXP_NO_DW_FORK=$v python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
done
