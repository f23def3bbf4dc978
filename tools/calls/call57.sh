cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v Warn | tail -12
import sys, time, argparse; sys.path.insert(0, ".")
import torch, bench
args = argparse.Namespace(seq=50, dtype="bf16", mode="faithful", batch=256, no_graph=True)
tr, step, _, _ = bench.build_mmgan(args, 0, torch.device("cuda", 0))
for _ in range(5): step()
torch.cuda.synchronize()
st0 = dict(torch.cuda.memory_stats())
ts = []
t0 = time.perf_counter()
for i in range(60):
    t = time.perf_counter(); step(); ts.append((time.perf_counter() - t) * 1e3)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
st1 = dict(torch.cuda.memory_stats())
print("host ms per step:", " ".join(f"{a:.2f}" for a in ts))
print("total host", (t1 - t0) * 1e3, "drain", (t2 - t1) * 1e3)
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams"):
    print(k, st0.get(k), "->", st1.get(k))
PY
