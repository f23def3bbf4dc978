cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GDM_LIB_TAG=stamps B=256 NB=1024 timeout -k 10 120 python tools/stamps.py c1 > gpurun_out/conv1_stamps.txt 2>&1; cat gpurun_out/conv1_stamps.txt
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_simnn_gpu.py -m gpu -x -q -k "conv1 or golden or trunk" 2>&1 | tail -2
