#!/usr/bin/env python3
"""Benchmark of the GAN training hot path on MI355X: training samples/s for one full G+D iteration.

    python bench.py --gpus N --steps K --warmup W [--workload simnn|mmgan] [--batch B] [--dtype bf16|fp32]
                    [--mode faithful|elided] [--no-cpu-baseline]

Default workload = BASELINE.json configs[1]: model 1 (GAN_DES/SIMNN.py) G+D iteration, 128x256 synthetic spectrogram
windows, batch 256 per GPU, bf16 MFMA, reference-faithful iteration (1 G forward, 3 D forwards, 2 D backwards, Adam).
N > 1: one process per GPU (torchrun env), per-GPU batch fixed (weak scaling), one RCCL all-reduce of the flat
discriminator-gradient bucket per iteration.  Rank 0 prints ONE JSON line.

Timing: W warm-up iterations, then exactly K iterations between barrier + device-synchronize brackets, wall clock,
MAX over ranks.  Inputs are resident in HBM before the timed region.  A second, instrumented pass of K iterations
brackets every launch of the dominant kernel with HIP events (on the launch stream) for the `roofline` object; the
CPU oracle (oracle/, this repo's restatement of the reference loop) is timed on the host cores for `cpu_baseline`
(rank 0, N == 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32 (spec; 155 measured)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 (MI355X_MICROARCH.md)

# dominant kernel (by total time in profiles/r01_*_kernel_stats.csv) per workload and its ALGORITHMIC bytes / flops per
# sample (derivation in DESIGN.md section 5)
def dominant_kernel(workload, dtype, hw, t):
    esz = 2 if dtype == "bf16" else 4
    if workload == "simnn":
        h, w = hw
        h1, w1 = (h + 1) // 2, (w + 1) // 2
        h2, w2 = h1 // 2, w1 // 2
        # fused conv2 data-gradient + conv1 weight-gradient kernel: reads the pooled gradient dp2 and its codes,
        # conv1's codes and the input window; writes nothing per sample (80 partial sums per workgroup)
        # (code2: one byte per channel PAIR since round 3)
        nbytes = h2 * w2 * 32 * esz + h2 * w2 * 16 + h1 * w1 * 8 + h * w * 4
        return {"name": "gdm_simnn_conv2_bwd_fused", "kernel": "conv2_bwd_data_kernel<FUSE> (gdm_simnn_conv2_bwd_fused)",
                "bytes_per_sample": nbytes, "flops_per_sample": 2.0 * h1 * w1 * 16 * 288}
    # model 2: the fused DiscriminatorCNN pass (forward + loss + backward of one sample inside LDS).  HBM sees the two
    # input planes only, so the kernel is priced against the bf16 MFMA peak: implicit-GEMM flops of conv1 / conv2
    # forward, conv2 dW, conv2 dX, conv1 dW and the two fc products (k4 s2 p1 convolutions, 2 -> 16 -> 32 channels)
    ow1 = t // 2
    ow2 = (ow1 - 2) // 2 + 1
    c1, c2, fc = 2.0 * 64 * ow1 * 16 * 32, 2.0 * 32 * ow2 * 32 * 256, 2.0 * 32 * 32 * ow2
    return {"name": "gdm_dcnn_fused", "kernel": "dcnn_fused_kernel (gdm_dcnn_fused)",
            "bytes_per_sample": 2 * 128 * t * 4, "flops_per_sample": 2 * c1 + 3 * c2 + 2 * fc}


def pipelined_flag(args):
    return args.workload == "simnn" and not args.no_pipeline and not args.no_overlap


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 / 20: a 20-step timing is dominated by what follows the barrier (first replay after an idle device: host launch
    # 0.3 ms instead of 0.12, clocks coming back up): 0.714 ms/step at 20 steps against 0.660 at 100 and 0.654 at 400 on
    # the same box.  The default run still takes well under a second of device time.
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="simnn", choices=["simnn", "mmgan"])
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--mode", default="faithful", choices=["faithful", "elided"])
    ap.add_argument("--width", type=int, default=256, help="spectrogram window width (simnn)")
    ap.add_argument("--seq", type=int, default=50, help="piano-roll length T (mmgan)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short secondary measurements (model 2 at B=256 / B=16, model 1 in exact fp32)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="model 1: run the generator half of an iteration inside the same call (SimnnTrainer.step)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-overlap", action="store_true", help="single stream (no concurrent branches)")
    ap.add_argument("--prime", type=int, default=40,
                    help="untimed set-up iterations BEFORE the W warm-up steps (device start-up transient, see measure())")
    ap.add_argument("--pieces", action="store_true",
                    help="model 1 on one rank: replay the five-graph form N > 1 ranks use (around the all-reduces)")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks ourselves (one process per GPU) as a
    CHILD `python -m torch.distributed.run ... bench.py <same arguments>`, relay its output (rank 0 prints the JSON
    line) and return its exit code.  This process never touches the GPU (no exec, no HIP call before or after)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on these hosts
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def setup_dist(n):
    from gan_des_midi_music_gen_amd import dp
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n > 1 or world > 1:
        if world != n:
            raise SystemExit(f"--gpus {n} but the torchrun environment has WORLD_SIZE={world}")
        return dp.init_from_env("nccl")
    torch.cuda.set_device(0)
    return 0, 1, 0


def rendezvous_only(args):
    """GDM_BENCH_RENDEZVOUS_ONLY=1 (CPU rehearsal of the launcher, tests/test_bench_launcher.py): every rank joins the
    process group (gloo), proves the collective works and rank 0 prints a JSON line -- no GPU is touched."""
    import torch.distributed as dist
    from gan_des_midi_music_gen_amd import dp
    os.environ.setdefault("GDM_DIST_BACKEND", "gloo")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ["GDM_DIST_BACKEND"])
    rank = dist.get_rank() if world > 1 else 0
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dp.allreduce_bucket_(t, 1)
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "requested": args.gpus,
                          "rank_sum": t.item()}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def build_simnn(args, rank, dev):
    from gan_des_midi_music_gen_amd import SIMNN, synthetic
    from gan_des_midi_music_gen_amd.train import SimnnTrainer
    hw = (128, args.width)
    torch.manual_seed(0)   # identical replicas on every rank
    gen = SIMNN.Generator().apply(SIMNN.weights_init).to(dev)
    disc = SIMNN.Discriminator(input_hw=hw).apply(SIMNN.weights_init).to(dev)
    tr = SimnnTrainer(gen, disc, compute_dtype=args.dtype, elide_dead_backward=(args.mode == "elided"),
                      overlap=not args.no_overlap)
    real, fake, noise = synthetic.simnn_inputs(args.batch, hw, seed=1234 + rank, device=dev)

    pipelined = not args.no_pipeline and not args.no_overlap

    def eager():
        return (tr.step_pipelined if pipelined else tr.step)(real, noise, fake)

    def eager_sequential():
        # the roofline pass: with the two halves of an iteration one after the other the dominant kernel has the chip
        # to itself, so the HIP events around its launches measure the kernel, not the time-sharing of two chains
        return tr.step(real, noise, fake)
    step = eager
    if not args.no_graph and (tr.world == 1 or (args.pieces and pipelined)):
        # one rank: the whole call as one hipGraph (+ the generator forward as a graph of its own).  N ranks launch
        # eagerly: the five-graph form around the two all-reduces (--pieces, SimnnTrainer._capture_pieces) is
        # bit-identical but measured SLOWER than the eager multi-stream iteration on one rank (0.755 vs 0.631 ms: the
        # graph boundaries and cross-stream joins cost more than ~60 host launches that the host issues ahead anyway)
        tr.capture(real, noise, fake, pipelined=pipelined, pieces=args.pieces)
        step = tr.replay
    elif pipelined:
        eager()        # every timed call must find a pending generator half, like every later one
    return tr, step, eager_sequential, eager


def build_mmgan(args, rank, dev):
    from gan_des_midi_music_gen_amd import network_tests as NT, synthetic
    from gan_des_midi_music_gen_amd.train import MmganTrainer
    torch.manual_seed(0)
    mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, args.seq), input_dim=50, output_dim=20,
                          instrument=0, start=100, end=100 + args.seq, device=dev)
    mm.train()
    tr = MmganTrainer(mm, compute_dtype=args.dtype, elide_dead_backward=(args.mode == "elided"))
    d = synthetic.mmgan_inputs(args.batch, args.seq, seed=1234 + rank, device=dev)

    def eager():
        return tr.step(d["piano_roll"], d["durations"], d["beats"], d["noise1"], d["noise2"], d["fake_a"],
                       d["fake_b"], g1_in_a=d["g1_in_a"], g1_in_b=d["g1_in_b"])
    step = eager
    if not args.no_graph:
        # one rank: the iteration is one hipGraph; N ranks: two graphs around the eager gradient all-reduce
        tr.capture(d["piano_roll"], d["durations"], d["beats"], d["noise1"], d["noise2"], d["fake_a"], d["fake_b"],
                   d["g1_in_a"], d["g1_in_b"])
        step = tr.replay
    return tr, step, eager, eager


def host_cores():
    """Threads for the CPU baseline: the affinity mask, clipped by the cgroup CPU quota and by GDM_CPU_THREADS
    (the GPU boxes expose 256 hardware threads but give one job a 16-CPU share; oversubscribing is 20x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("GDM_CPU_THREADS", "16"))))


def cpu_baseline(args):
    """Oracle (CPU restatement of the reference loop) on the host cores; bounded sample, same geometry/mode.

    Returns (cpu_baseline object, parity record): the oracle starts from the benchmark's weights (seed 0) and inputs
    (seed 1234), so the (disc_loss, gen_loss) pairs of its iterations -- the warm-up call is iteration 0 -- are what a
    fresh GPU trainer on the same state must reproduce (``loss_parity``).  parity = {"init": state_dicts before the
    first iteration, "losses": [(d, g), ...]}."""
    import copy
    from oracle import simnn as osn, mmgan as om, steps as ost
    from gan_des_midi_music_gen_amd import synthetic
    cores = host_cores()
    torch.set_num_threads(cores)
    elide = args.mode == "elided"
    losses = []
    if args.workload == "simnn":
        b, hw = args.batch, (128, args.width)      # the benchmarked batch (256: ~4 s per iteration on 16 threads)
        torch.manual_seed(0)
        gen = osn.Generator().apply(osn.weights_init)
        disc = osn.Discriminator(input_hw=hw).apply(osn.weights_init)
        g_opt = ost.Adam(gen.parameters(), lr=0.00002, betas=(0.5, 0.999))
        d_opt = ost.Adam(disc.parameters(), lr=0.00002, betas=(0.5, 0.999))
        real, fake, noise = synthetic.simnn_inputs(b, hw, seed=1234)
        init = (copy.deepcopy(gen.state_dict()), copy.deepcopy(disc.state_dict()))
        fn = lambda: losses.append(ost.simnn_iteration(gen, disc, g_opt, d_opt, real, noise, fake, elide)[:2])  # noqa: E731
        sample = f"oracle.simnn_iteration, batch {b}, 128x{args.width}, fp32, {args.mode}"
    else:
        b = args.batch
        torch.manual_seed(0)
        mm = om.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, args.seq), input_dim=50, output_dim=20)
        g_opt = ost.Adam(list(mm.generator1.parameters()) + list(mm.generator2.parameters()), lr=0.01)
        d_opt = ost.Adam(mm.discriminator.parameters(), lr=0.01)
        d = synthetic.mmgan_inputs(b, args.seq, seed=1234)
        init = (copy.deepcopy(mm.state_dict()),)
        fn = lambda: losses.append(ost.mmgan_iteration(mm, g_opt, d_opt, d["piano_roll"], d["durations"],  # noqa: E731
                                                       d["beats"], d["noise1"], d["noise2"], d["g1_in_a"], d["g1_in_b"],
                                                       d["fake_a"], d["fake_b"], 1, elide)[:2])
        sample = f"oracle.mmgan_iteration, batch {b}, T={args.seq}, fp32, {args.mode}"
    fn()                                           # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if (el > 12.0 and n >= 3) or n >= 200 or el > 40.0:      # >= 3 iterations, ~12-20 s of CPU work
            break
    return ({"value": round(b * n / el, 2), "unit": "samples/s", "cores": cores, "kind": "port",
             "sample": f"{sample}; {n} iterations in {el:.1f} s"}, {"init": init, "losses": losses})


def loss_parity(args, parity, dev):
    """The metric's "D-loss parity vs CPU": a FRESH trainer (the oracle's initial weights, the benchmark inputs, the
    benchmarked arithmetic) runs as many iterations as the oracle did for ``cpu_baseline`` and its (disc_loss, gen_loss)
    pairs are compared with the oracle's (GAN_DES/SIMNN.py:289-334 / network_tests.py:304-321 restated in
    oracle/steps.py).  Model 1 runs free (tolerance = the parity tests' 50-iteration bound); model 2's bf16 trajectory is
    only comparable from identical state, so only its first iteration is held to the bound."""
    from gan_des_midi_music_gen_amd import SIMNN, network_tests as NT, synthetic
    from gan_des_midi_music_gen_amd.train import MmganTrainer, SimnnTrainer
    want = parity["losses"]
    elide = args.mode == "elided"
    got = []
    if args.workload == "simnn":
        hw = (128, args.width)
        gen, disc = SIMNN.Generator(), SIMNN.Discriminator(input_hw=hw)
        gen.load_state_dict(parity["init"][0])
        disc.load_state_dict(parity["init"][1])
        tr = SimnnTrainer(gen.to(dev), disc.to(dev), compute_dtype=args.dtype, elide_dead_backward=elide)
        real, fake, noise = synthetic.simnn_inputs(args.batch, hw, seed=1234, device=dev)
        for _ in want:
            tr.step(real, noise, fake)
            got.append((tr.disc_loss_value(), tr.gen_loss_value()))
        k = len(want)
        tol = 2e-3 if args.dtype == "bf16" else 1e-4
        rel = False
    else:
        mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, args.seq), input_dim=50, output_dim=20,
                              instrument=0, start=100, end=100 + args.seq, device=dev)
        mm.load_state_dict(parity["init"][0])
        mm.to(dev).train()
        tr = MmganTrainer(mm, compute_dtype=args.dtype, elide_dead_backward=elide)
        d = synthetic.mmgan_inputs(args.batch, args.seq, seed=1234, device=dev)
        tr.step(d["piano_roll"], d["durations"], d["beats"], d["noise1"], d["noise2"], d["fake_a"], d["fake_b"],
                g1_in_a=d["g1_in_a"], g1_in_b=d["g1_in_b"])
        got.append((tr.disc_loss_value(), tr.gen_loss_value()))
        k = 1
        tol = 2e-2 if args.dtype == "bf16" else 2e-3
        rel = True
    dd = [abs(g[0] - w[0]) / (max(1.0, abs(w[0])) if rel else 1.0) for g, w in zip(got[:k], want[:k])]
    dg = [abs(g[1] - w[1]) / (max(1.0, abs(w[1])) if rel else 1.0) for g, w in zip(got[:k], want[:k])]
    ok = all(x == x and x <= tol for x in dd + dg)
    del tr
    return {"iters": k, "max_abs_d": float(f"{max(dd):.3e}"), "max_abs_g": float(f"{max(dg):.3e}"), "tol": tol,
            "relative_to_max_1_abs_loss": rel, "pass": bool(ok),
            "gpu": [[round(a, 6), round(b, 6)] for a, b in got[:k]],
            "cpu": [[round(a, 6), round(b, 6)] for a, b in want[:k]],
            "what": "fresh trainer vs oracle/steps.py from the same weights (seed 0) and inputs (seed 1234), "
                    + ("free-running" if not rel else "first iteration (model 2 is compared from identical state only)")}


def launch_description(args, world, tr):
    if args.no_graph or (world > 1 and args.workload == "simnn" and getattr(tr, "_pieces", None) is None):
        return "eager (multi-stream)"
    if getattr(tr, "_pieces", None) is not None:
        return "5 fork-free hipGraphs (generator, generator half, forward+head+fc1 dW, backward, Adam) around 2 eager all-reduces"
    if world > 1:
        return "2 hipGraphs + eager all-reduce per iteration"
    if getattr(tr, "_graph_gen", None) is not None:
        return "hipGraph replay: main graph + the generator forward as a graph on a stream of its own"
    return "hipGraph replay"


def measure(args, rank, world, dev, barrier, dist):
    """Warm up, time exactly args.steps iterations between barriers, then (optionally) the per-launch timing passes of
    the dominant kernel.  Returns a dict."""
    import gc
    from gan_des_midi_music_gen_amd import ops
    tr, step, eager_step, step_eager_same_schedule = (build_simnn if args.workload == "simnn" else build_mmgan)(args, rank, dev)
    # A generation-2 pass of Python's garbage collector walks every object `import torch` created: a 20-45 ms pause of
    # ONE host call (found with GDM_BENCH_STEP_TIMES=1: one step() of an eager 20-step run took 43.7 ms, the others
    # 0.35), i.e. 1-2 ms per step of a 20-step timing, at random.  Collect BEFORE the warm-up and keep the collector out
    # of the warm-up and the timed region: the timed region then follows the warm-up's barrier directly, with the device
    # still warm (a collection between the two left the device idle for the 20-45 ms it takes).
    gc.collect()
    gc.disable()
    try:
        # Device start-up transient (tools/experiments/replay_transient.py, profiles/r03_replay_transient.txt): coming from
        # an idle device, the first replay takes 1.09 ms, replays 4-10 0.70-0.75 ms, and the iteration only settles at its
        # steady 0.61 ms after ~30-40 replays (~25 ms of work: clocks and the relative phase of the two replay streams).
        # A 5-step warm-up + 20 timed steps measures that transient (0.66 ms), not the training rate.  `--prime` untimed
        # iterations (default 40, reported in the JSON line) run as part of set-up, then the W warm-up steps, then
        # exactly K timed steps between the barriers.
        for _ in range(max(0, args.prime)):
            step()
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        _dbg = os.environ.get("GDM_BENCH_STEP_TIMES")
        _ts = []
        for _ in range(args.steps):
            _t = time.perf_counter()
            step()
            if _dbg:
                _ts.append((time.perf_counter() - _t) * 1e3)
        if _dbg:
            _t = time.perf_counter()
        barrier()
        elapsed = time.perf_counter() - t0
    finally:
        gc.enable()
    if _dbg:
        print("host ms per step:", " ".join(f"{a:.2f}" for a in _ts), " drain", (time.perf_counter() - _t) * 1e3, file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if hasattr(tr, "flush"):
        tr.flush()       # pipelined schedule: the last iteration's generator half (its twin ran before the timed region)
    losses = (tr.disc_loss_value(), tr.gen_loss_global())      # global-batch means (gen_loss: one tiny collective)

    roofline = None
    if not args.no_roofline:
        dk = dominant_kernel(args.workload, args.dtype, (128, args.width), args.seq)

        def time_kernel(fn):
            """K more iterations, eagerly (a replayed graph has no host calls to bracket), with every launch of the
            dominant kernel between two HIP events recorded on the stream it is launched on."""
            ops.time_entry_point(dk["name"])
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            res = ops.timed_durations_ms()
            ops.time_entry_point(None)
            return res
        # (a) alone: the halves of an iteration one after the other, so the kernel has the chip to itself; (b) in the
        # schedule that was timed above (model 1: pipelined, the other chain runs beside it and the launch is
        # time-shared).  `frac` is taken on the SLOWER of the two.
        alone_ms, launches = time_kernel(eager_step)
        sched_ms, _ = time_kernel(step_eager_same_schedule) if step_eager_same_schedule is not eager_step else (alone_ms, 0)
        avg_ms = max(alone_ms, sched_ms)
        if hasattr(tr, "flush"):
            tr.flush()
        if args.workload == "simnn":
            # the timed entry point runs on the 2B batch (D step) and, in faithful mode, on B (the dead backward)
            samples_per_launch = 2.0 * args.batch if args.mode == "elided" else 1.5 * args.batch
            if args.dtype == "bf16":
                achieved = dk["bytes_per_sample"] * samples_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
                roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": dk["kernel"],
                            "algorithmic_bytes_per_launch": int(dk["bytes_per_sample"] * samples_per_launch)}
            else:
                # exact-fp32 mode: 192 FLOP per algorithmic byte against a ridge of 157 TF / 8 TB/s = 20 FLOP/B -> the
                # binding roof is the fp32 MFMA
                tflops = dk["flops_per_sample"] * samples_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
                roofline = {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(tflops / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None,
                            "kernel": dk["kernel"],
                            "algorithmic_flops_per_launch": int(dk["flops_per_sample"] * samples_per_launch)}
            roofline.update({"avg_launch_ms": round(avg_ms, 4), "avg_launch_ms_alone": round(alone_ms, 4),
                             "avg_launch_ms_in_timed_schedule": round(sched_ms, 4), "launches_timed": launches})
        else:
            # the timed entry point runs on the 2B batch (D step) and on B (the generator step's pass through D)
            samples_per_launch = 1.5 * args.batch
            tflops = dk["flops_per_sample"] * samples_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            roofline = {"bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None, "kernel": dk["kernel"],
                        "algorithmic_flops_per_launch": int(dk["flops_per_sample"] * samples_per_launch),
                        "avg_launch_ms": round(avg_ms, 4), "avg_launch_ms_alone": round(alone_ms, 4),
                        "avg_launch_ms_in_timed_schedule": round(sched_ms, 4), "launches_timed": launches}
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if roofline and os.path.exists(tfile):
            try:     # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh): a
                roofline["traffic"] = json.load(open(tfile)).get(f"{args.workload}_{args.dtype}")     # separate run
                roofline["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc, separate passes, same command)"
            except Exception:
                pass
    barrier()
    res = {"elapsed": elapsed, "losses": losses, "roofline": roofline, "launch": launch_description(args, world, tr)}
    del tr, step, eager_step, step_eager_same_schedule
    return res


def secondary_lines(args, rank, dev, barrier):
    """Short measurements of the other configurations the driver's single command should see (one rank only): model 2
    (BASELINE configs[2]) at B = 256 and at the reference's batch 16, and model 1 in the reference's own precision
    (exact fp32).  Each: own warm-up, 50 timed replays, own roofline object."""
    import copy
    out = {}
    for name, over in (("mmgan_b256_bf16", dict(workload="mmgan", batch=256, dtype="bf16")),
                       ("mmgan_b16_bf16", dict(workload="mmgan", batch=16, dtype="bf16")),
                       ("simnn_b256_fp32", dict(workload="simnn", batch=256, dtype="fp32"))):
        if over["workload"] == args.workload and over["batch"] == args.batch and over["dtype"] == args.dtype:
            continue
        a = copy.copy(args)
        for k, v in over.items():
            setattr(a, k, v)
        a.steps, a.warmup, a.prime = 50, 10, 20
        torch.cuda.empty_cache()
        r = measure(a, rank, 1, dev, barrier, None)
        out[name] = {"ms_per_step": round(1e3 * r["elapsed"] / a.steps, 4),
                     "value": round(a.batch * a.steps / r["elapsed"], 2), "unit": "samples/s", "steps": a.steps,
                     "warmup": a.warmup, "per_gpu_batch": a.batch, "dtype": "bf16" if a.dtype == "bf16" else "f32",
                     "launch": r["launch"], "roofline": r["roofline"],
                     "final_losses": {"disc": round(r["losses"][0], 6), "gen": round(r["losses"][1], 6)}}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))        # before anything touches the GPU in this process
    if os.environ.get("GDM_BENCH_RENDEZVOUS_ONLY") == "1":
        return rendezvous_only(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    rank, world, local = setup_dist(args.gpus)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from gan_des_midi_music_gen_amd import _lib
    if _lib.load().gdm_build_flavor() != 0 and os.environ.get("GDM_BENCH_ALLOW_EXPERIMENT") != "1":   # (A/B runs of variant libraries)
        raise SystemExit("libgdm_hip.so was built with experiment switches (GDM_HIPCC_FLAGS): rebuild with the shipped "
                         "flags (`python -m gan_des_midi_music_gen_amd.build`) before benchmarking")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = measure(args, rank, world, dev, barrier, dist)
    elapsed, losses, roofline = res["elapsed"], res["losses"], res["roofline"]
    rc = 0
    if rank == 0:
        total = world * args.batch * args.steps
        out = {
            "metric": "GAN training samples/sec (G+D step)", "value": round(total / elapsed, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "priming_steps": max(0, args.prime),
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {
                "workload": ("SIMNN G/D iteration, synthetic 128x%d spectrogram windows" % args.width
                             if args.workload == "simnn" else
                             "MMGAN (G + beat-G + D) iteration, MAESTRO-shaped synthetic (2,128,%d) rolls" % args.seq),
                "per_gpu_batch": args.batch, "global_batch": args.batch * world, "mode": args.mode,
                "parallelism": f"dp{world}", "launch": res["launch"],
                "iteration": "1 G fwd, 3 D fwd, 2 D bwd, Adam(D)" if
                args.workload == "simnn" else "2x(G1,G2) fwd, 3 D fwd, 2 D bwd, Adam(D)",
                **({"schedule": "pipelined: each call = D step of iteration i + generator half (D pass on fake, label 1) "
                                "of iteration i-1 on a side stream; K timed calls do K of each; results bit-identical "
                                "to the sequential schedule"}
                   if pipelined_flag(args) else {}),
            },
            "final_losses": {"disc": round(losses[0], 6), "gen": round(losses[1], 6)},
        }
        if roofline is not None:
            out["roofline"] = roofline
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            st = json.load(open(tfile)).get(f"{args.workload}_{args.dtype}_step")
            if st:
                # whole-iteration HBM bytes (rocprofv3 --pmc, separate passes) over SURVEY.md 8d's algorithmic bytes
                out["step_traffic_ratio"] = {"measured_bytes_per_step": st["measured_bytes_per_step"],
                                             "algorithmic_bytes_per_step": st["algorithmic_bytes_per_step"],
                                             "ratio": round(st["measured_bytes_per_step"] / st["algorithmic_bytes_per_step"], 3),
                                             "source": st.get("source", "profiles/traffic.json")}
        except Exception:
            pass
        if world == 1 and not args.no_secondary:
            out["secondary"] = secondary_lines(args, rank, dev, barrier)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], parity = cpu_baseline(args)
            out["loss_parity"] = loss_parity(args, parity, dev)
            if not out["loss_parity"]["pass"]:
                rc = 3
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
