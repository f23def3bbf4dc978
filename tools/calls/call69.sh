cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --batch 16 --width 64 --no-cpu-baseline --no-roofline > gpurun_out/bench_simnn_c1_b16_w64.json 2>/dev/null; cut -c1-330 gpurun_out/bench_simnn_c1_b16_w64.json
python bench.py --batch 16 --width 216 --no-cpu-baseline --no-roofline > gpurun_out/bench_simnn_c1_b16_w216.json 2>/dev/null; cut -c100-175 gpurun_out/bench_simnn_c1_b16_w216.json
python bench.py --batch 128 --width 216 --no-cpu-baseline --no-roofline > gpurun_out/bench_simnn_c5_b128_w216.json 2>/dev/null; cut -c100-175 gpurun_out/bench_simnn_c5_b128_w216.json
