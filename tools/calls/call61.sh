cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c100-170; python bench.py --no-cpu-baseline --no-roofline --steps 100 --warmup 20 2>/dev/null | cut -c100-170; python bench.py --no-cpu-baseline --no-roofline --steps 400 --warmup 50 2>/dev/null | cut -c100-170; done
