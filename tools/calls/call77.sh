cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GDM_BENCH_ALLOW_EXPERIMENT=1
run() { echo -n "$1 $2: "; env $1 python bench.py $2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do run GDM_LIB_TAG= ""; run GDM_LIB_TAG=nt4 ""; done
for i in 1 2 3; do run GDM_LIB_TAG= "--workload mmgan"; run GDM_LIB_TAG=nt4 "--workload mmgan"; done
