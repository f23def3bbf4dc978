set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in 1 0; do echo "PAIR=$v"; GDM_EXP_C1_PAIR=$v python bench.py --no-cpu-baseline --no-secondary --no-roofline | cut -c100-200; done; done
