cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
run X=0
run GDM_BD_CAP=384
run GDM_BD_CAP=448
run GDM_BD_CAP=640
run GDM_BW_CAP=512
run GDM_BW_CAP=640
run GDM_BW_CAP=1024
run GDM_C2F_CAP=512
run GDM_C2F_CAP=640
run GDM_C2F_CAP=1024
run GDM_C1_CAP=1024
run GDM_C1_CAP=1536
run GDM_C1_CAP=3072
run X=1
