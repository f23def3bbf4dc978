# kernel stats of the single-stream eager step after the Adam fusion (why did the step get slower?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/p4_seq
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_seq -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-graph --no-overlap > gpurun_out/p4_seq.log 2>&1
python tools/step_breakdown.py gpurun_out/p4_seq > gpurun_out/p4_seq_breakdown.txt 2>&1
python tools/step_breakdown.py gpurun_out/p4_seq v > gpurun_out/p4_seq_timeline.txt 2>&1
python bench.py --no-cpu-baseline --no-roofline > gpurun_out/p4_graph.json 2>&1
python bench.py --no-cpu-baseline --no-roofline --no-pipeline > gpurun_out/p4_graph_nopipe.json 2>&1
cat gpurun_out/p4_seq_breakdown.txt; grep -h metric gpurun_out/p4_seq.log gpurun_out/p4_graph.json gpurun_out/p4_graph_nopipe.json | cut -c1-180
find gpurun_out/p4_seq -name "*.db" -delete
