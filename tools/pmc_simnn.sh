cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT

rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmc_a -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --prime 0 --no-roofline > gpurun_out/pmc_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --prime 0 --no-roofline > gpurun_out/pmc_b.log 2>&1
ls gpurun_out/pmc_a/*/ gpurun_out/pmc_b/*/
