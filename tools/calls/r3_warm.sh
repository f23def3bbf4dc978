cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "20 5" "20 20" "20 50" "40 5" "100 5" "200 20"; do set -- $cfg; echo "steps=$1 warmup=$2: $(python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-secondary --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"; done
GDM_BENCH_STEP_TIMES=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | grep "host ms"
