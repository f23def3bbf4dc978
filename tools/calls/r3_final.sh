bash tools/calls/r3_round.sh
python tools/graph_timeline.py gpurun_out/final_simnn > gpurun_out/simnn_replay_timeline.txt 2>&1
python tools/graph_timeline.py gpurun_out/final_mmgan > gpurun_out/mmgan_replay_timeline.txt 2>&1
GDM_LIB_TAG= python tools/experiments/adam_ab.py > gpurun_out/adam_ab.txt 2>&1
