"""GPU, 2 ranks on ONE device over gloo (RCCL refuses two ranks on one GPU): the data-parallel trainer path end to end
-- per-rank shards, early asynchronous all-reduce of fc1's gradient, head all-reduce, 1/world in Adam -- must
reproduce the single-process iteration on the global batch (fp32 mode: parameters to ~1e-6, losses to 1e-5)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["GDM_ROOT"])
from gan_des_midi_music_gen_amd import SIMNN, dp, synthetic, network_tests as NT
from gan_des_midi_music_gen_amd.train import SimnnTrainer, MmganTrainer
rank, world, devi = dp.init_from_env()
dev = torch.device("cuda", devi)
GB, hw = 4, (32, 40)
torch.manual_seed(0)
gen = SIMNN.Generator().apply(SIMNN.weights_init).to(dev)
disc = SIMNN.Discriminator(input_hw=hw).apply(SIMNN.weights_init).to(dev)
tr = SimnnTrainer(gen, disc, compute_dtype="fp32")
lo, hi = dp.shard_bounds(GB, world, rank)
out = {}
for it in range(3):
    real, fake, noise = synthetic.simnn_inputs(GB, hw, seed=900 + it, device=dev)
    dl, gl = tr.step(real[lo:hi].contiguous(), noise[lo:hi].contiguous(), fake[lo:hi].contiguous())
torch.cuda.synchronize()
out["simnn"] = {"d_loss": tr.disc_loss_value(), "g_loss": tr.gen_loss_global(), "fc1": disc.fc1.weight.detach().cpu(), "c1": disc.conv1.weight.detach().cpu(),
                "fc2b": disc.fc2.bias.detach().cpu()}
torch.manual_seed(0)
gen = SIMNN.Generator().apply(SIMNN.weights_init).to(dev)
disc = SIMNN.Discriminator(input_hw=hw).apply(SIMNN.weights_init).to(dev)
trp = SimnnTrainer(gen, disc, compute_dtype="fp32")
for it in range(3):
    real, fake, noise = synthetic.simnn_inputs(GB, hw, seed=900 + it, device=dev)
    trp.step_pipelined(real[lo:hi].contiguous(), noise[lo:hi].contiguous(), fake[lo:hi].contiguous())
torch.cuda.synchronize()
g_mid = trp.gen_loss_global()      # gen_loss of the iteration before the last one: written BEFORE the last exchange
trp.flush()
torch.cuda.synchronize()
out["simnn_pipelined"] = {"d_loss": trp.disc_loss_value(), "g_loss": trp.gen_loss_global(), "g_loss_mid": g_mid, "fc1": disc.fc1.weight.detach().cpu(),
                          "c1": disc.conv1.weight.detach().cpu(), "fc2b": disc.fc2.bias.detach().cpu()}
torch.manual_seed(0)
mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, 50), input_dim=50, output_dim=20, device=dev)
mt = MmganTrainer(mm, compute_dtype="fp32")
for it in range(2):
    d = synthetic.mmgan_inputs(GB, 50, seed=950 + it, device=dev)
    sl = slice(lo, hi)
    mt.step(d["piano_roll"][sl].contiguous(), d["durations"][sl].contiguous(), d["beats"][sl].contiguous(),
            d["noise1"][sl].contiguous(), d["noise2"][sl].contiguous(), d["fake_a"][sl].contiguous(),
            d["fake_b"][sl].contiguous(), g1_in_a=d["g1_in_a"][sl].contiguous(), g1_in_b=d["g1_in_b"][sl].contiguous())
torch.cuda.synchronize()
out["mmgan"] = {"d_loss": mt.disc_loss_value(), "g_loss": mt.gen_loss_global(), "fc": mm.discriminator.fc.weight.detach().cpu(),
                "c2": mm.discriminator.conv2.weight.detach().cpu()}
# model 2 on the benchmarked path (bf16, fused kernels): eager steps vs the replayed graph(s) -- with 2 ranks that is two
# hipGraphs around the eager gradient all-reduce
d = synthetic.mmgan_inputs(GB, 50, seed=960, device=dev)
sh = {k: v[lo:hi].contiguous() for k, v in d.items()}
keys = ("piano_roll", "durations", "beats", "noise1", "noise2", "fake_a", "fake_b", "g1_in_a", "g1_in_b")
for kind in ("eager", "graph"):
    torch.manual_seed(0)
    mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, 50), input_dim=50, output_dim=20, device=dev)
    mt = MmganTrainer(mm, compute_dtype="bf16")
    if kind == "graph":
        g = mt.capture(*[sh[k] for k in keys])              # two eager iterations inside
        assert isinstance(g, tuple) == (world > 1)
        for _ in range(2):
            mt.replay()
    else:
        for _ in range(4):
            mt.step(*[sh[k] for k in keys[:7]], g1_in_a=sh["g1_in_a"], g1_in_b=sh["g1_in_b"])
    torch.cuda.synchronize()
    out["mmgan_bf16_" + kind] = {"d_loss": mt.disc_loss_value(), "g_loss": mt.gen_loss_global(),
                                 "fc": mm.discriminator.fc.weight.detach().cpu(),
                                 "c1": mm.discriminator.conv1.weight.detach().cpu()}
# exact global-batch BatchNorm statistics for the generators (SURVEY.md 8e "exact mode"): with 2 ranks the generated
# matrices / parameters and the running statistics must equal ONE process on the whole batch (world = 1: same code path off)
torch.manual_seed(0)
mm = NT.MultiModalGAN(z_dim=50, adj_size=(64, 64), roll_size=(2, 128, 50), input_dim=50, output_dim=20, device=dev)
me = MmganTrainer(mm, compute_dtype="fp32", exact_bn=True)
for it in range(2):
    d = synthetic.mmgan_inputs(GB, 50, seed=980 + it, device=dev)
    sl = slice(lo, hi)
    me.step(d["piano_roll"][sl].contiguous(), d["durations"][sl].contiguous(), d["beats"][sl].contiguous(),
            d["noise1"][sl].contiguous(), d["noise2"][sl].contiguous(), d["fake_a"][sl].contiguous(),
            d["fake_b"][sl].contiguous(), g1_in_a=d["g1_in_a"][sl].contiguous(), g1_in_b=d["g1_in_b"][sl].contiguous())
torch.cuda.synchronize()
g1_all = dp.all_gather_cat(me.last_g1.contiguous()) if world > 1 else me.last_g1
g2_all = dp.all_gather_cat(me.last_g2.contiguous()) if world > 1 else me.last_g2
out["mmgan_exact_bn"] = {"g1": g1_all.detach().cpu(), "g2": g2_all.detach().cpu(),
                         "rv": mm.generator1.gen[3][1].running_var.detach().cpu(),
                         "rm": mm.generator2.gen[0][1].running_mean.detach().cpu(),
                         "nbt": int(mm.generator1.gen[0][1].num_batches_tracked.item())}
# model 1 on the benchmarked path (bf16, pipelined schedule): eager calls vs the five-graph replay around the two eager
# collectives (train.SimnnTrainer._capture_pieces) -- what N > 1 ranks run in bench.py
real, fake, noise = (t[lo:hi].contiguous() for t in synthetic.simnn_inputs(GB, hw, seed=970, device=dev))
for kind in ("eager", "pieces"):
    torch.manual_seed(0)
    gen = SIMNN.Generator().apply(SIMNN.weights_init).to(dev)
    disc = SIMNN.Discriminator(input_hw=hw).apply(SIMNN.weights_init).to(dev)
    tg = SimnnTrainer(gen, disc, compute_dtype="bf16")
    if kind == "pieces":
        gg = tg.capture(real, noise, fake, pipelined=True, pieces=True)     # two eager iterations inside
        assert isinstance(gg, dict) and len(gg) == 5
        for _ in range(3):
            tg.replay()
    else:
        for _ in range(5):
            tg.step_pipelined(real, noise, fake)
    torch.cuda.synchronize()
    g_mid = tg.gen_loss_global()
    tg.flush()
    torch.cuda.synchronize()
    out["simnn_bf16_" + kind] = {"d_loss": tg.disc_loss_value(), "g_loss": tg.gen_loss_global(), "g_loss_mid": g_mid,
                                 "fc1": disc.fc1.weight.detach().cpu(), "c1": disc.conv1.weight.detach().cpu(),
                                 "gen": tg.last_generated.detach().cpu(), "bn": gen.batch_norm2.running_var.detach().cpu()}
if rank == 0:
    torch.save(out, os.environ["GDM_OUT"])
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, out_path, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, GDM_ROOT=ROOT, GDM_OUT=str(out_path), GDM_DIST_BACKEND="gloo", GDM_SINGLE_DEVICE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world))
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]


def test_two_ranks_equal_one_process_on_the_global_batch(tmp_path):
    _run(1, tmp_path / "one.pt", tmp_path)
    _run(2, tmp_path / "two.pt", tmp_path)
    one = torch.load(tmp_path / "one.pt", weights_only=True)
    two = torch.load(tmp_path / "two.pt", weights_only=True)
    # the pipelined schedule is the same arithmetic: identical to the sequential one within a run
    for k in ("fc1", "c1", "fc2b"):
        assert torch.equal(one["simnn"][k], one["simnn_pipelined"][k]), k
    # graph replay (one graph on one rank, two graphs around the eager all-reduce on two) == eager steps, bit for bit
    for run in (one, two):
        e, g = run["mmgan_bf16_eager"], run["mmgan_bf16_graph"]
        assert e["d_loss"] == g["d_loss"] and e["g_loss"] == g["g_loss"], (e["d_loss"], g["d_loss"])
        assert torch.equal(e["fc"], g["fc"]) and torch.equal(e["c1"], g["c1"])
        e, g = run["simnn_bf16_eager"], run["simnn_bf16_pieces"]
        assert e["d_loss"] == g["d_loss"] and e["g_loss"] == g["g_loss"] and e["g_loss_mid"] == g["g_loss_mid"]
        for k in ("fc1", "c1", "gen", "bn"):
            assert torch.equal(e[k], g[k]), ("model 1 pieces replay vs eager", k)
    # exact global-batch BatchNorm: two ranks' generators see the statistics of the whole batch
    a, b = one["mmgan_exact_bn"], two["mmgan_exact_bn"]
    assert a["nbt"] == b["nbt"] == 4
    for k in ("g1", "g2", "rv", "rm"):
        assert (a[k] - b[k]).abs().max().item() < 2e-5 * max(1.0, a[k].abs().max().item()), ("exact_bn", k)
    for model in ("simnn", "simnn_pipelined", "mmgan"):
        a, b = one[model], two[model]
        assert abs(a["d_loss"] - b["d_loss"]) < 1e-5 * max(1.0, abs(a["d_loss"])), (model, a["d_loss"], b["d_loss"])
        for k in [k for k in a if k.startswith("g_loss")]:
            # gen_loss is rank-local state (never summed by the gradient exchange, also not under the pipelined
            # schedule, which writes it before the next exchange); its global mean equals the single process's
            assert abs(a[k] - b[k]) < 1e-5 * max(1.0, abs(a[k])), (model, k, a[k], b[k])
        for k in a:
            if not k.endswith("_loss") and not k.startswith("g_loss"):
                # Adam's first steps move every weight by ~lr regardless of |g|: allow a few 1e-6 of drift
                assert (a[k] - b[k]).abs().max().item() < 3e-5, (model, k, (a[k] - b[k]).abs().max().item())
