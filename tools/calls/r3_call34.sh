cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do for v in 0 1; do echo -n "PAIR=$v "; GDM_EXP_C1_PAIR=$v python bench.py --no-cpu-baseline --no-secondary --no-roofline | grep -o '"ms_per_step": [0-9.]*'; done; done
