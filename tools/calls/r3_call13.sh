set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r3_t13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_t13.log
tail -8 gpurun_out/r3_t13.log
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-230 gpurun_out/bench_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline | cut -c1-230
