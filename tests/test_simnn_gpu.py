"""GPU: model 1 (GAN_DES/SIMNN.py surface) against the golden vectors captured from the reference and against the
CPU oracle on seeded inputs.

Stated tolerances: fp32 mode -- outputs rtol 1e-5 (of the output scale), losses |d| <= 1e-5,
gradients rel-L2 <= 1e-4 (fc1 has K = 55 296 terms per dot product), 10 free-running iterations |dloss| <= 1e-4;
bf16 mode -- outputs/losses within 2e-2, gradients rel-L2 <= 5e-2.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gan_des_midi_music_gen_amd import SIMNN, functional as Fn, optim, synthetic  # noqa: E402
from gan_des_midi_music_gen_amd.train import SimnnTrainer  # noqa: E402
from oracle import simnn as osn, steps as ost  # noqa: E402  (checker only)

from helpers import (assert_summary_close, load_golden, record, rel_l2, round_gradient, round_operand,  # noqa: E402
                     tensor_summary, weight_digest)

DEV = "cuda"


def _build(seed, interleaved, input_hw=(128, 216)):
    torch.manual_seed(seed)
    if interleaved:
        gen = SIMNN.Generator().apply(SIMNN.weights_init)
        disc = SIMNN.Discriminator(input_hw=input_hw).apply(SIMNN.weights_init)
    else:
        gen, disc = SIMNN.Generator(), SIMNN.Discriminator(input_hw=input_hw)
        gen.apply(SIMNN.weights_init)
        disc.apply(SIMNN.weights_init)
    return gen, disc


def _close(got, want, rtol, what=""):
    got = torch.as_tensor(got).detach().float().cpu()
    want = torch.as_tensor(want).detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def test_same_seed_gives_reference_weights():
    g = load_golden("simnn_modules.npz")
    gen, disc = _build(int(g["seed"]), False)
    for name, mod in (("gen", gen), ("disc", disc)):
        for k, v in mod.state_dict().items():
            assert weight_digest(v) == g[f"digest/{name}/{k}"], f"{name}.{k}"


def test_modules_match_golden_fp32():
    g = load_golden("simnn_modules.npz")
    gen, disc = _build(int(g["seed"]), False)
    gen.to(DEV), disc.to(DEV)
    real, fake, noise = (torch.from_numpy(g[k]).to(DEV) for k in ("real", "fake", "noise"))
    gen.train()
    out = gen(noise)
    assert out.shape == (2, 1, 20, 20)
    _close(out, g["gen_out_train"], 1e-5, "G train output")
    for k, v in gen.state_dict().items():
        if "running" in k or "num_batches" in k:
            _close(v, g[f"gen_after_fwd/{k}"], 2e-5, k)
    gen.eval()
    with torch.no_grad():
        _close(gen(noise), g["gen_out_eval"], 1e-5, "G eval output")
    # generator backward (module completeness; the reference's loop never calls it)
    gen2, _ = _build(int(g["seed"]), True)
    gen2.to(DEV).train()
    nz = noise.clone().requires_grad_(True)
    (gen2(nz) * torch.from_numpy(g["gen_bwd_R"]).to(DEV)).sum().backward()
    for k, p in gen2.named_parameters():
        assert_summary_close(tensor_summary(p.grad), g[f"gen_grad/{k}"], 2e-4, 1e-9, k)
    assert rel_l2(nz.grad, g["gen_grad/noise"]) < 2e-4
    # discriminator forward + the two BCE terms + backward through autograd
    d_real = disc(real)
    _close(d_real, g["disc_out_real"], 1e-5, "D(real)")
    b = real.shape[0]
    l_real = F.binary_cross_entropy_with_logits(d_real.reshape(-1), torch.full((b,), 0.9, device=DEV))
    l_fake = F.binary_cross_entropy_with_logits(disc(fake).reshape(-1), torch.full((b,), 0.1, device=DEV))
    assert abs(l_real.item() - float(g["loss_real"])) < 1e-5
    assert abs(l_fake.item() - float(g["loss_fake"])) < 1e-5
    (l_real + l_fake).backward()
    for k, p in disc.named_parameters():
        assert_summary_close(tensor_summary(p.grad), g[f"disc_grad/{k}"], 2e-4, 1e-10, k)
    for k in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "fc1.bias", "fc2.weight", "fc2.bias"):
        assert rel_l2(dict(disc.named_parameters())[k].grad, g[f"disc_grad_full/{k}"]) < 1e-4, k


def test_generator_checkpoint_of_the_reference_loads_strict():
    ck = load_golden("simnn_gen_ckpt.npz")
    gen = SIMNN.Generator()
    gen.load_state_dict({k[3:]: torch.from_numpy(ck[k]) for k in ck.files if k.startswith("sd/")}, strict=True)
    gen.to(DEV).eval()
    with torch.no_grad():
        out = gen(torch.from_numpy(ck["noise"]).to(DEV))
    _close(out, ck["gen_out_eval"], 1e-5, "checkpoint generator output")


def test_trainer_reproduces_golden_iterations_fp32():
    g = load_golden("simnn_steps.npz")
    b = int(g["batch"])
    results = {}
    for elide in (False, True):
        gen, disc = _build(int(g["seed"]), True)
        gen.to(DEV), disc.to(DEV)
        tr = SimnnTrainer(gen, disc, lr=0.00002, betas=(0.5, 0.999), compute_dtype="fp32", elide_dead_backward=elide)
        dls, gls = [], []
        for it in range(10):
            real, fake, noise = synthetic.simnn_inputs(b, (128, 216), seed=100 + it, device=DEV)
            dl, gl = tr.step(real, noise, fake)
            dls.append(dl.item())
            gls.append(gl.item())
            if it == 0:
                _close(tr.last_generated, g["generated_it1"], 1e-5, "generated matrices")
            if not elide and it + 1 in (1, 2, 10):
                np.testing.assert_allclose(disc.conv1.weight.detach().cpu().numpy(),
                                           g[f"disc_after_{it + 1}_full/conv1.weight"], rtol=0, atol=5e-6)
                np.testing.assert_allclose(disc.fc2.weight.detach().cpu().numpy(),
                                           g[f"disc_after_{it + 1}_full/fc2.weight"], rtol=0, atol=5e-6)
                for k, v in gen.state_dict().items():
                    assert_summary_close(tensor_summary(v.float()), g[f"gen_after_{it + 1}/{k}"], 2e-5, 1e-8, k)
                for k, v in disc.state_dict().items():
                    want = g[f"disc_after_{it + 1}/{k}"]
                    assert abs(tensor_summary(v.float())[1] - want[1]) <= 1e-5 * abs(want[1]) + 1e-7, k
        np.testing.assert_allclose(dls, g["disc_losses"], rtol=0, atol=1e-4)
        np.testing.assert_allclose(gls, g["gen_losses"], rtol=0, atol=1e-4)
        assert abs(dls[0] - g["disc_losses"][0]) < 1e-5 and abs(gls[0] - g["gen_losses"][0]) < 1e-5
        results[elide] = (dls, gls, disc.fc1.weight.detach().clone())
        assert all(p.grad is None for p in gen.parameters())
    # eliding the dead backward changes nothing observable, bit for bit
    assert results[False][0] == results[True][0] and results[False][1] == results[True][1]
    assert torch.equal(results[False][2], results[True][2])


@pytest.mark.parametrize("mode,tol_out,tol_grad", [("fp32", 2e-5, 2e-4), ("bf16", 2e-2, 5e-2)])
def test_discriminator_vs_oracle_at_benchmark_geometry(mode, tol_out, tol_grad):
    hw = (128, 256)
    torch.manual_seed(3)
    ref = osn.Discriminator(input_hw=hw).apply(osn.weights_init)
    disc = SIMNN.Discriminator(input_hw=hw)
    disc.load_state_dict(ref.state_dict(), strict=True)
    disc.to(DEV)
    disc.compute_dtype = mode
    x = synthetic.spectrogram_batch(3, hw, seed=11)
    p_ref = ref(x)
    l_ref = ost.bce_with_logits(p_ref.reshape(-1), torch.full((3,), 0.9))
    l_ref.backward()
    p = disc(x.to(DEV))
    _close(p, p_ref, tol_out, f"D output {mode}")
    loss = F.binary_cross_entropy_with_logits(p.reshape(-1), torch.full((3,), 0.9, device=DEV))
    loss.backward()
    assert abs(loss.item() - l_ref.item()) < (1e-5 if mode == "fp32" else 2e-2)
    for (k, pr), (_, pg) in zip(ref.named_parameters(), disc.named_parameters()):
        assert rel_l2(pg.grad, pr.grad) < tol_grad, (k, rel_l2(pg.grad, pr.grad))


def test_module_level_loop_with_fused_adam_matches_trainer():
    """The reference's loop written with the drop-in modules + optim.Adam equals the fused trainer (fp32)."""
    hw = (128, 216)
    b = 2
    outs = []
    for use_trainer in (False, True):
        gen, disc = _build(5, True)
        gen.to(DEV), disc.to(DEV)
        if use_trainer:
            tr = SimnnTrainer(gen, disc, compute_dtype="fp32")
        else:
            gen_opt = optim.Adam(gen.parameters(), lr=0.00002, betas=(0.5, 0.999))
            disc_opt = optim.Adam(disc.parameters(), lr=0.00002, betas=(0.5, 0.999))
        losses = []
        for it in range(3):
            real, fake, noise = synthetic.simnn_inputs(b, hw, seed=300 + it, device=DEV)
            if use_trainer:
                dl, gl = tr.step(real, noise, fake)
                losses.append((dl.item(), gl.item()))
                continue
            crit = torch.nn.BCEWithLogitsLoss()
            disc_opt.zero_grad()
            l_real = crit(disc(real).reshape(-1), torch.ones(b, device=DEV) * 0.9)
            _generated = gen(noise)
            l_fake = crit(disc(fake.detach()).reshape(-1), torch.ones(b, device=DEV) * 0.1)
            d_loss = l_fake + l_real
            d_loss.backward()
            disc_opt.step()
            gen_opt.zero_grad()
            g_loss = crit(disc(fake).squeeze(), torch.ones(b, device=DEV))
            g_loss.backward()
            gen_opt.step()
            losses.append((d_loss.item(), g_loss.item()))
        outs.append((losses, disc.fc2.weight.detach().cpu().clone(), gen.batch_norm1.running_mean.cpu().clone()))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs[0][1].numpy(), outs[1][1].numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs[0][2].numpy(), outs[1][2].numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_full_size_properties(mode):
    """BASELINE config 1 size (B=256, 128x256): batch independence, gradient additivity, run-to-run determinism."""
    hw, b = (128, 256), 256
    torch.manual_seed(1)
    disc = SIMNN.Discriminator(input_hw=hw).apply(SIMNN.weights_init).to(DEV)
    disc.compute_dtype = mode
    x = synthetic.spectrogram_batch(b, hw, seed=5, device=DEV)
    with torch.no_grad():
        full = disc(x)
        parts = torch.cat([disc(x[:100]), disc(x[100:])])
        again = disc(x)
    assert torch.equal(full, again), "forward must be deterministic"
    assert (full - parts).abs().max().item() <= (1e-5 if mode == "fp32" else 2e-3), "samples must be independent"
    assert torch.isfinite(full).all() and full.min() >= 0 and full.max() <= 1

    def grads(sl):
        disc.zero_grad()
        p = disc(x[sl])
        p.backward(torch.ones_like(p) / b)
        return [q.grad.detach().clone() for q in disc.parameters()]

    g_all, g_a, g_b = grads(slice(0, b)), grads(slice(0, 128)), grads(slice(128, b))
    for ga, g1, g2, (k, _) in zip(g_all, g_a, g_b, disc.named_parameters()):
        assert rel_l2(ga, g1 + g2) < (1e-4 if mode == "fp32" else 2e-2), k
    g_again = grads(slice(0, b))
    assert all(torch.equal(u, v) for u, v in zip(g_all, g_again)), "backward must be deterministic"


def test_stream_overlap_and_graph_replay_are_bit_identical_to_single_stream():
    hw, b = (128, 216), 4
    outs = []
    for mode in ("single", "overlap", "graph"):
        gen, disc = _build(7, True)
        gen.to(DEV), disc.to(DEV)
        tr = SimnnTrainer(gen, disc, compute_dtype="bf16", overlap=(mode != "single"))
        real, fake, noise = synthetic.simnn_inputs(b, hw, seed=55, device=DEV)
        if mode == "graph":
            tr.capture(real, noise, fake)           # runs 2 warm-up iterations itself
            for _ in range(3):
                dl, gl = tr.replay()
        else:
            for _ in range(5):
                dl, gl = tr.step(real, noise, fake)
        torch.cuda.synchronize()
        outs.append((dl.item(), gl.item(), disc.fc1.weight.detach().clone(), disc.conv1.weight.detach().clone(),
                     gen.batch_norm2.running_var.clone(), int(gen.batch_norm1.num_batches_tracked.item())))
    for mode, o in zip(("overlap", "graph"), outs[1:]):
        assert o[0] == outs[0][0] and o[1] == outs[0][1], (mode, o[:2], outs[0][:2])
        assert torch.equal(o[2], outs[0][2]) and torch.equal(o[3], outs[0][3]) and torch.equal(o[4], outs[0][4]), mode
        assert o[5] == outs[0][5] == 5, mode


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_pipelined_iterations_are_bit_identical_to_sequential(dtype):
    """step_pipelined runs the generator half of iteration i beside the discriminator step of i+1: losses (gen_loss
    one call later), parameters, Adam state and generator BN statistics must equal the sequential step's bit for
    bit -- eagerly, with varying inputs, and as a replayed hipGraph."""
    hw, b, n = (32, 40), 4, 5
    batches = [synthetic.simnn_inputs(b, hw, seed=70 + i, device=DEV) for i in range(n)]

    def run(kind):
        gen, disc = _build(9, True, input_hw=hw)
        gen.to(DEV), disc.to(DEV)
        tr = SimnnTrainer(gen, disc, compute_dtype=dtype)
        dls, gls = [], []
        if kind == "seq":
            for real, fake, noise in batches:
                dl, gl = tr.step(real, noise, fake)
                dls.append(dl.item()); gls.append(gl.item())
        else:
            for i, (real, fake, noise) in enumerate(batches):
                dl, gl = tr.step_pipelined(real, noise, fake)
                dls.append(dl.item())
                if i > 0:
                    gls.append(gl.item())
            gls.append(tr.flush().item())
        torch.cuda.synchronize()
        return dls, gls, [v.detach().clone() for v in tr.d.views], gen.batch_norm3.running_mean.clone(), tr

    seq, pipe = run("seq"), run("pipe")
    assert seq[0] == pipe[0], (seq[0], pipe[0])
    assert seq[1] == pipe[1], (seq[1], pipe[1])
    assert all(torch.equal(u, v) for u, v in zip(seq[2], pipe[2]))
    assert torch.equal(seq[3], pipe[3])
    # mixing: a sequential step after pipelined ones flushes the pending half first
    tr = pipe[4]
    real, fake, noise = batches[0]
    tr.step_pipelined(real, noise, fake)
    tr.step(real, noise, fake)
    assert tr._pending_fake is None


@pytest.mark.parametrize("dtype,b,hw", [("bf16", 4, (128, 216)), ("fp32", 4, (32, 40)), ("bf16", 260, (32, 40))])
def test_pipelined_graph_replay_matches_sequential_steps(dtype, b, hw):
    """The main graph and the generator graph replay CONCURRENTLY on two streams.  In fp32 mode (layer-wise generator:
    split-K GEMM, bn_act_fwd) and for B > 256 (bn_stats) the generator graph uses scratch buffers too: each graph must
    have its own (ops.workspace_namespace), or the two race -- losses, weights, generated matrices and the generator's
    BatchNorm statistics are compared bit for bit with sequential eager steps."""
    outs = []
    for mode in ("seq", "graph"):
        gen, disc = _build(7, True, input_hw=hw)
        gen.to(DEV), disc.to(DEV)
        tr = SimnnTrainer(gen, disc, compute_dtype=dtype)
        real, fake, noise = synthetic.simnn_inputs(b, hw, seed=55, device=DEV)
        if mode == "graph":
            tr.capture(real, noise, fake, pipelined=True)       # 2 warm-up calls run; the captured call does not
            assert tr._graph_gen is not None
            for _ in range(3):
                dl, _ = tr.replay()
            gl = tr.flush()
        else:
            for _ in range(5):
                dl, gl = tr.step(real, noise, fake)
        torch.cuda.synchronize()
        outs.append((dl.item(), gl.item(), disc.fc1.weight.detach().clone(), disc.conv2.weight.detach().clone(),
                     tr.last_generated.clone(), gen.batch_norm1.running_var.clone(), gen.batch_norm3.running_mean.clone()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1], (outs[0][:2], outs[1][:2])
    for k in range(2, 7):
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_pipelined_replay_with_refilled_inputs_matches_eager_steps():
    """The generator runs as a graph of its own on the trainer's stream: a caller that refills the static input buffers
    between replays (what a data loader does) must get, per iteration, exactly what eager steps on the same inputs give
    -- generated matrices included (they read ``noise`` on the other stream)."""
    hw, b, n = (128, 216), 4, 6
    batches = [synthetic.simnn_inputs(b, hw, seed=300 + i, device=DEV) for i in range(n)]
    runs = []
    for mode in ("seq", "graph"):
        gen, disc = _build(9, True)
        gen.to(DEV), disc.to(DEV)
        tr = SimnnTrainer(gen, disc, compute_dtype="bf16")
        gens = []
        if mode == "graph":
            real, fake, noise = (t.clone() for t in batches[0])
            tr.capture(real, noise, fake, pipelined=True)            # 2 warm-up iterations on batch 0
            for i in range(2, n):
                for dst, src in zip((real, fake, noise), batches[i]):
                    dst.copy_(src)                                     # no synchronisation: stream order must do
                tr.replay()
                gens.append(tr.last_generated.clone())
            tr.flush()
        else:
            for i in (0, 0) + tuple(range(2, n)):
                tr.step_pipelined(*[batches[i][k] for k in (0, 2, 1)])
                if i >= 2:
                    gens.append(tr.last_generated.clone())
            tr.flush()
        torch.cuda.synchronize()
        runs.append((gens, disc.fc1.weight.detach().clone(), tr.disc_loss_value(), tr.gen_loss_value()))
    for a, b_ in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b_)
    assert torch.equal(runs[0][1], runs[1][1]) and runs[0][2:] == runs[1][2:]


def test_train_entry_point_runs_and_checkpoints(tmp_path):
    gen, disc, g_losses, d_losses = SIMNN.train(None, batch_size=4, max_steps=7, model_path=str(tmp_path), seed=0,
                                                log=lambda *_: None)
    assert len(g_losses) == 7 and len(d_losses) == 7 and all(np.isfinite(g_losses)) and all(np.isfinite(d_losses))
    saved = list(tmp_path.glob("gen_5_*.pt"))
    assert len(saved) == 1
    sd = torch.load(saved[0], weights_only=True)
    assert set(sd) == set(SIMNN.Generator().state_dict())


def test_geometry_mismatch_is_a_clear_error():
    disc = SIMNN.Discriminator().to(DEV)
    with pytest.raises(ValueError):
        disc(torch.zeros(2, 128, 256, device=DEV))


def test_discriminator_input_gradient_matches_golden_fp32():
    """x.requires_grad_(): the module is an ordinary autograd citizen like the reference's (SIMNN.py:129-142)."""
    g = load_golden("input_grads.npz")
    _, disc = _build(0, False)
    disc.to(DEV)
    x = torch.from_numpy(g["simnn/x"]).to(DEV).requires_grad_(True)
    b = x.shape[0]
    F.binary_cross_entropy_with_logits(disc(x).reshape(-1), torch.full((b,), 0.9, device=DEV)).backward()
    assert x.grad.shape == x.shape
    assert rel_l2(x.grad, g["simnn/x_grad"]) < 2e-4, rel_l2(x.grad, g["simnn/x_grad"])
    assert rel_l2(disc.conv1.weight.grad, g["simnn/conv1_weight_grad"]) < 2e-4
    # bf16 activations.  The input gradient is a POINTWISE quantity (<= 64 terms per pixel, no averaging over the batch
    # like the weight gradients' 5e-2), behind two bf16-stored gradient maps and cancelling sums over 128 / 288 terms:
    # measured rel-L2 8.4e-2, bound 1.5e-1
    disc.compute_dtype = "bf16"
    x2 = torch.from_numpy(g["simnn/x"]).to(DEV).requires_grad_(True)
    F.binary_cross_entropy_with_logits(disc(x2).reshape(-1), torch.full((b,), 0.9, device=DEV)).backward()
    q = rel_l2(x2.grad, g["simnn/x_grad"])
    assert q < 1.5e-1, q
    # What of that is the KERNELS' error and what is bf16 quantisation: the same network on the CPU with every value
    # the bf16 path rounds rounded at the same place -- p1, p2, conv2 / fc1 / fc2 weight operands, h1 as fc2's operand
    # (forward); dz, dh1 as GEMM operands and the stored gradient maps dp2, dp1 (backward) -- and everything else in
    # fp32.  Against THAT the input gradient has to agree to 1e-4 (summation order; measured 1.3e-7).
    cpu = {k: v.detach().cpu() for k, v in disc.state_dict().items()}
    xr = torch.from_numpy(g["simnn/x"]).requires_grad_(True)
    a1 = F.max_pool2d(torch.relu(F.conv2d(xr.unsqueeze(1), cpu["conv1.weight"], cpu["conv1.bias"], padding=1)), 2, 2)
    a1 = round_gradient(round_operand(a1))                                    # p1 / dp1 stored in bf16
    a2 = F.max_pool2d(torch.relu(F.conv2d(a1, round_operand(cpu["conv2.weight"]), cpu["conv2.bias"], padding=1)), 2, 2)
    a2 = round_gradient(round_operand(a2))                                    # p2 / dp2 stored in bf16
    z1 = round_gradient(F.linear(a2.flatten(1), round_operand(cpu["fc1.weight"]), cpu["fc1.bias"]))   # dh1: bf16 operand
    h1 = torch.relu(z1)
    z = round_gradient(F.linear(round_operand(h1), round_operand(cpu["fc2.weight"]), cpu["fc2.bias"]))  # dz: bf16 operand
    F.binary_cross_entropy_with_logits(torch.sigmoid(z).reshape(-1), torch.full((b,), 0.9)).backward()
    qk = rel_l2(x2.grad, xr.grad)
    record("simnn_input_gradient_bf16", vs_fp32_golden=q, vs_same_rounding_cpu=qk)
    assert qk < 1e-4, qk


def test_simnn_net_matches_golden_and_reference_shape_test():
    """SimNN (GAN_DES/SIMNN.py:145-170).  Golden: same constructor seed -> same conv/fc2 weights (digests), same seed
    right before forward -> same re-created fc1 -> same five outputs.  Then the reference's own ad-hoc test
    (test_SimNN, SIMNN.py:218-231: output shapes for random input sizes) at sizes that fit."""
    g = load_golden("simnn_net.npz")
    torch.manual_seed(7)
    net = SIMNN.SimNN(6)
    for k, v in net.state_dict().items():
        assert weight_digest(v) == g[f"digest/{k}"], k
    net.to(DEV)
    x = torch.from_numpy(g["x"]).to(DEV)
    torch.manual_seed(123)
    outs = net(x)
    assert weight_digest(net.fc1.weight) == g["fc1_weight_digest"] and net.fc1.in_features == int(g["fc1_in_features"])
    for name, o in zip(("matrix", "array1", "array2", "array3", "array4"), outs):
        _close(o, g[name], 2e-5, f"SimNN {name}")
    # backward (module completeness) against torch CPU autograd on the same weights
    ref = torch.nn.Sequential()       # plain fp32 restatement of the op chain
    xc = torch.from_numpy(g["x"]).requires_grad_(True)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    y = F.max_pool2d(F.relu(F.conv2d(xc, sd["conv1.weight"], sd["conv1.bias"], padding=1)), 2, 2)
    y = F.max_pool2d(F.relu(F.conv2d(y, sd["conv2.weight"], sd["conv2.bias"], padding=1)), 2, 2)
    w_ref = [sd[k].clone().requires_grad_(True) for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")]
    out_ref = F.linear(F.relu(F.linear(y.flatten(1), w_ref[0], w_ref[1])), w_ref[2], w_ref[3])
    R = torch.randn(out_ref.shape, generator=torch.Generator().manual_seed(3))
    (out_ref * R).sum().backward()
    xg = x.clone().requires_grad_(True)
    fc1 = net.fc1
    out = Fn.SimnnNetFn.apply(xg, net.conv1.weight, net.conv1.bias, net.conv2.weight, net.conv2.bias, fc1.weight,
                              fc1.bias, net.fc2.weight, net.fc2.bias, Fn.F32)
    (out * R.to(DEV)).sum().backward()
    assert rel_l2(xg.grad, xc.grad) < 2e-4
    assert rel_l2(fc1.weight.grad, w_ref[0].grad) < 2e-4 and rel_l2(net.fc2.bias.grad, w_ref[3].grad) < 2e-4
    # the reference's shape test, feasible sizes (it draws sizes up to 32768^2: not runnable anywhere)
    n, bs = 10, 4
    model = SIMNN.SimNN(n).to(DEV)
    gsz = torch.Generator().manual_seed(0)
    for _ in range(5):
        size = int(torch.randint(16, 200, (1,), generator=gsz).item())
        matrix, a1, a2, a3, a4 = model(torch.randn(bs, 1, size, size, device=DEV))
        assert matrix.size() == (bs, n, n)
        assert all(a.size() == (bs, n) for a in (a1, a2, a3, a4))


def test_pipelined_step_owns_its_copy_of_the_fake_batch():
    """A loader / bridge that refills its output buffer in place (pinned staging buffers, pre-rendered window pools):
    step_pipelined keeps its own copy for the generator half that runs one call later, so overwriting ``fake`` right
    after the call returns must not change anything -- eagerly and under graph replay."""
    hw, b, n = (32, 40), 4, 4
    batches = [synthetic.simnn_inputs(b, hw, seed=170 + i, device=DEV) for i in range(n)]

    def run(mutate, graph):
        gen, disc = _build(11, True, input_hw=hw)
        gen.to(DEV), disc.to(DEV)
        tr = SimnnTrainer(gen, disc, compute_dtype="bf16")
        real, fake, noise = (t.clone() for t in batches[0])
        if graph:
            tr.capture(real, noise, fake, pipelined=True)
        gls = []
        for i in range(n):
            for dst, src in zip((real, fake, noise), batches[i]):
                dst.copy_(src)
            _, gl = tr.replay() if graph else tr.step_pipelined(real, noise, fake)
            gls.append(gl.item())
            if mutate:
                fake.fill_(123.0)              # the caller reuses its buffer before the pending half has run
        gls.append(tr.flush().item())
        torch.cuda.synchronize()
        return gls, disc.fc1.weight.detach().clone(), disc.conv1.weight.detach().clone()

    for graph in (False, True):
        clean, dirty = run(False, graph), run(True, graph)
        assert clean[0] == dirty[0], (graph, clean[0], dirty[0])
        assert torch.equal(clean[1], dirty[1]) and torch.equal(clean[2], dirty[2]), graph


def test_workspace_buffers_seen_by_a_graph_are_never_released():
    """ops.workspace: a buffer handed out during a capture stays alive when a later, larger request replaces it (the
    graph has its address baked in); replay after such a growth still gives the captured result."""
    from gan_des_midi_music_gen_amd import ops
    x = torch.randn(64, 4096, device=DEV)
    w = torch.randn(4096, 64, device=DEV)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        want = ops.gemm(x, w, split_k=8).clone()          # warm-up on the capture stream's key
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = ops.gemm(x, w, split_k=8)                    # split-K slabs live in the stream's workspace
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    retired_before = len(ops._ws_retired)
    with torch.cuda.stream(side):
        big = ops.gemm(torch.randn(512, 8192, device=DEV), torch.randn(8192, 512, device=DEV), split_k=32)   # grows it
    torch.cuda.synchronize()
    assert len(ops._ws_retired) == retired_before + 1, "the superseded buffer must be retired, not freed"
    junk = [torch.full((1 << 20,), 7.0, device=DEV) for _ in range(8)]     # would land in freed memory
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want) and big.isfinite().all() and len(junk) == 8


@pytest.mark.parametrize("b", [2, 5, 13, 256])
def test_fused_generator_forward_matches_the_layerwise_path_and_the_oracle(b):
    """The trainers' forward-only generator (csrc/simnn_gen.hip: BatchNorm+ReLU applied while the next transposed
    convolution stages its input, per-workgroup BatchNorm partials) against (a) the layer-by-layer GEMM + col2im +
    batch-norm path on the same bf16 operands and (b) the fp32 CPU oracle, incl. batches that do not fill the last
    workgroup's sample group."""
    torch.manual_seed(31)
    ref = osn.Generator().apply(osn.weights_init)
    gens = []
    for _ in range(2):
        g = SIMNN.Generator()
        g.load_state_dict(ref.state_dict())
        gens.append(g.to(DEV).train())
    noise = torch.randn(b, 100, 1, 1, generator=torch.Generator().manual_seed(b))
    outs = []
    for g, need_bwd in zip(gens, (False, True)):
        ws = [g.conv1.weight.detach(), g.conv2.weight.detach(), g.conv3.weight.detach(), g.conv4.weight.detach()]
        bns = [(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.num_batches_tracked)
               for bn in (g.batch_norm1, g.batch_norm2, g.batch_norm3)]
        img, saved = Fn.simnn_gen_forward(noise.to(DEV), ws, bns, True, Fn.BF16, cache={}, need_backward=need_bwd)
        assert (saved is None) == (not need_bwd)
        outs.append(img)
    want = ref.train()(noise)
    assert outs[0].shape == (b, 1, 20, 20)
    _close(outs[0], outs[1], 2e-4, "fused vs layer-wise generator")
    _close(outs[0], want, 1e-3, "fused generator vs oracle")
    for k in ("batch_norm1", "batch_norm2", "batch_norm3"):
        fa, fb, fr = getattr(gens[0], k), getattr(gens[1], k), getattr(ref, k)
        assert int(fa.num_batches_tracked) == int(fb.num_batches_tracked) == 1
        _close(fa.running_mean, fb.running_mean, 1e-3, k + " running_mean vs layer-wise")
        _close(fa.running_var, fb.running_var, 1e-3, k + " running_var vs layer-wise")
        _close(fa.running_mean, fr.running_mean, 1e-2, k + " running_mean vs oracle")      # bf16 operands, tiny batches
        _close(fa.running_var, fr.running_var, 1e-2, k + " running_var vs oracle")


def test_non_finite_values_are_reported_like_anomaly_mode():
    """The reference's loops run under torch.autograd.set_detect_anomaly(True) (network_tests.py:211) and propagate NaN
    into the losses.  Here the arithmetic has no NaN semantics (-fno-honor-nans), so the defined behaviour is: under
    anomaly mode a step raises ops.NonFiniteError for a NaN input or an Inf weight; without it ``check_finite`` reports
    them on request; the autograd Functions of the module path raise under anomaly mode as well."""
    from gan_des_midi_music_gen_amd import ops
    hw, b = (32, 40), 4
    gen, disc = _build(3, True, input_hw=hw)
    gen.to(DEV), disc.to(DEV)
    tr = SimnnTrainer(gen, disc, compute_dtype="bf16")
    real, fake, noise = synthetic.simnn_inputs(b, hw, seed=1, device=DEV)
    tr.step(real, noise, fake)
    tr.check_finite(real, fake, noise)                          # clean state: no error
    bad = real.clone()
    bad[1, 5, 7] = float("nan")
    with torch.autograd.detect_anomaly(check_nan=False):
        with pytest.raises(ops.NonFiniteError):
            tr.step(bad, noise, fake)
    tr.step(real, noise, fake)                                  # anomaly mode off: no check, no exception ...
    with pytest.raises(ops.NonFiniteError):
        tr.check_finite(bad)                                    # ... unless asked, with the batch in hand
    # an Inf weight: found in the flat parameter buffer, with or without a NaN ever reaching the loss
    gen2, disc2 = _build(3, True, input_hw=hw)
    gen2.to(DEV), disc2.to(DEV)
    tr2 = SimnnTrainer(gen2, disc2, compute_dtype="bf16")
    with torch.no_grad():
        disc2.fc1.weight[3, 11] = float("inf")
    tr2.invalidate_weights()
    tr2.step(real, noise, fake)
    with pytest.raises(ops.NonFiniteError):
        tr2.check_finite()
    # module path: the custom Functions look at their tensors under anomaly mode
    gen3, disc3 = _build(3, True, input_hw=hw)
    disc3.to(DEV)
    with torch.autograd.detect_anomaly(check_nan=False):
        out = disc3(real)                                       # clean input: fine
        out.sum().backward()
        with pytest.raises(ops.NonFiniteError):
            disc3(bad)
    assert int(ops.nonfinite_count([bad, real.bfloat16(), torch.tensor([float("-inf")], device=DEV)]).item()) == 2


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_one_launch_optimizer_step_equals_the_four_launch_chain(dtype):
    """disc_opt.step() (GAN_DES/SIMNN.py:316) as ONE launch (gdm_simnn_adam_step: bias-correction terms, Adam on the
    small parameters, transposing Adam on fc1.weight, conv2 re-pack) against adam_prep -> Adam -> Adam_pc -> re-pack:
    parameters, moments, the packed conv2 images, fc1's operand copy, losses and the device step counter bit for bit."""
    hw, b = (32, 40), 4
    runs = []
    for one in (True, False):
        gen, disc = _build(11, True, input_hw=hw)
        gen.to(DEV), disc.to(DEV)
        tr = SimnnTrainer(gen, disc, compute_dtype=dtype, one_launch_optimizer=one)
        losses = []
        for it in range(4):
            real, fake, noise = synthetic.simnn_inputs(b, hw, seed=40 + it, device=DEV)
            dl, gl = tr.step(real, noise, fake)
            losses.append((dl.item(), gl.item()))
            if it == 1:
                tr.lr = 1e-5                       # a hyper-parameter change between steps (device record rewritten)
        torch.cuda.synchronize()
        runs.append((losses, tr.d.flat.clone(), tr.d.exp_avg.clone(), tr.d.exp_avg_sq.clone(), tr._prepared[0].clone(),
                     tr._prepared[1].clone(), int(tr.d._hyper.view(torch.int32)[0].item()), tr.d.step_count))
    a, c = runs
    assert a[0] == c[0], (a[0], c[0])
    for k in range(1, 6):
        assert torch.equal(a[k], c[k]), k
    assert a[6] == c[6] == 4 and a[7] == c[7] == 4
