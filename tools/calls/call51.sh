cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/experiments/iterations_per_graph.py 2>&1 | grep -v Warn | tail -4
