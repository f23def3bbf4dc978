cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 1; do echo "SPLIT_ADAM=$v"; GDM_EXP_SPLIT_ADAM=$v python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-220; done
done
