cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v Warn | tail -12
import sys, time, argparse; sys.path.insert(0, ".")
import torch, bench
args = argparse.Namespace(seq=50, dtype="bf16", mode="faithful", batch=256, no_graph=True)
tr, step, _, _ = bench.build_mmgan(args, 0, torch.device("cuda", 0))
ts = []
for i in range(40):
    torch.cuda.synchronize(); t = time.perf_counter()
    step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t) * 1e3, (t2 - t) * 1e3))
print(" ".join(f"{a:.2f}/{b:.2f}" for a, b in ts))
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"])
PY
