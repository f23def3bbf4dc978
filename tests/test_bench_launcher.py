"""CPU: `python bench.py --gpus N` starts N ranks by itself (child torchrun, one process per GPU) -- the form the
driver uses for the scaling runs.  Rehearsed with GDM_BENCH_RENDEZVOUS_ONLY=1: the ranks join a gloo process group,
all-reduce their rank numbers and rank 0 prints a JSON line; no GPU is touched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GDM_BENCH_RENDEZVOUS_ONLY="1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup",
                        "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_ranks_that_rendezvous():
    out = _run(2)
    assert out["n_gpus"] == 2 and out["requested"] == 2 and out["rank_sum"] == 3.0


def test_bench_gpus_1_runs_in_process():
    out = _run(1)
    assert out["n_gpus"] == 1 and out["rank_sum"] == 1.0


def test_launcher_precedes_any_gpu_call():
    """The parent must not initialise HIP before (or after) spawning: the launch is the first statement of main()."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(args.gpus)") < main.index("torch.cuda")
