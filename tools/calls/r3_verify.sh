cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/verify_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/verify_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/verify_bench.json 2> gpurun_out/verify_bench.err; echo "bench rc=$?"; cut -c1-220 gpurun_out/verify_bench.json
