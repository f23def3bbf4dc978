# PMC counters of the conv kernels alone (tools/bench_op.py at B=512), two passes of 8 SQ counters each.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d gpurun_out/pmc_oa -- python tools/bench_op.py > gpurun_out/pmc_oa.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_ob -- python tools/bench_op.py > gpurun_out/pmc_ob.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_oa gpurun_out/pmc_ob > gpurun_out/pmc_ops_summary.txt 2>&1
