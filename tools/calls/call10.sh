mkdir -p gpurun_out; L=gpurun_out/r2_streams.log; : > $L
python tools/stream_concurrency.py >> $L 2>&1
GPU_MAX_HW_QUEUES=8 python tools/stream_concurrency.py >> $L 2>&1
GPU_MAX_HW_QUEUES=2 python tools/stream_concurrency.py >> $L 2>&1
GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --no-roofline >> $L 2>&1
GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --no-roofline --no-graph >> $L 2>&1
GPU_MAX_HW_QUEUES=8 python tools/bench_conv_overlap.py >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-200
