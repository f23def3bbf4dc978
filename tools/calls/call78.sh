cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.Stream.priority_range())"
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2; do run GDM_EXP_GEN_PRIO=0; run GDM_EXP_GEN_PRIO=1; run GDM_EXP_GEN_PRIO=-1; done
