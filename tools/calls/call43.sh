cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -5
python bench.py 2>/dev/null | cut -c1-400
