cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do run X=0; run GDM_EXP_MAIN_PRIO=-1; done
run() { echo -n "$1 mmgan: "; env $1 python bench.py --workload mmgan --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do run X=0; run GDM_EXP_MAIN_PRIO=-1; done
