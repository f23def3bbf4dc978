cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for tag in head ""; do echo "lib=${tag:-new}"; GDM_LIB_TAG=$tag python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
done
