cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/overlap_probe_mmgan.py 2>&1 | grep -v Warn | tail -4
B=16 python tools/overlap_probe_mmgan.py 2>&1 | grep -v Warn | tail -4
