cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "simnn or ops" 2>&1 | tail -2
for rep in 1 2; do
for tag in head ""; do echo "lib=${tag:-new}"; GDM_LIB_TAG=$tag python tools/bench_op.py 2>&1 | grep -v Warn | grep "conv2_fwd"; GDM_LIB_TAG=$tag python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | cut -c90-170; done
done
