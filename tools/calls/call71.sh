cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "simnn or ops or trainer_parity or dp or abi" 2>&1 | tail -3
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c100-175; done
