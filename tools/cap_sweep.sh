# Grid-cap sweep of the persistent conv kernels under the default (pipelined, graph-replayed) bench: two chains share the
# chip, so the caps tuned for a kernel running alone are worth re-checking.  Prints ms/step per setting.
run() { echo -n "$1: "; env $1 python bench.py --no-cpu-baseline --no-secondary --no-roofline --steps 40 --warmup 5 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
run X=0
run GDM_BD_CAP=384
run GDM_BD_CAP=768
run GDM_BW_CAP=512
run GDM_BW_CAP=1024
run GDM_C2F_CAP=512
run GDM_C2F_CAP=1024
run GDM_C1_CAP=1024
run GDM_C1_CAP=4096
run X=1
