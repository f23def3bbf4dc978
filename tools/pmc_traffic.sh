# HBM traffic of every kernel of the model-1 step: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --prime 0 --no-roofline --no-graph --no-overlap > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --prime 0 --no-roofline --no-graph --no-overlap > gpurun_out/pmc_write.log 2>&1
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/
