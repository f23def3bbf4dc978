set -x
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_trunk" > gpurun_out/r3_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t2.log; tail -30 gpurun_out/r3_t2.log
