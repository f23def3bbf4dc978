cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do run GDM_C1_CAP=2048; run GDM_C1_CAP=1536; run GDM_C1_CAP=1280; done
