"""CPU oracle (test infrastructure only): loss, optimiser, scheduler and the two training-iteration bodies.

Restates, with explicit formulas on fp32 CPU tensors:
  bce_with_logits  nn.BCEWithLogitsLoss() (mean) as used at GAN_DES/SIMNN.py:257,289,311,329 and
                   MMGAN_MIDI_DES/network_tests.py:248,304-305,313
  Adam             torch.optim.Adam (no amsgrad, no weight decay) as configured at SIMNN.py:258-259
                   (lr 2e-5, betas (0.5, 0.999)) and network_tests.py:253-254 (lr 0.01, defaults)
  StepLR           torch.optim.lr_scheduler.StepLR(step_size=30, gamma=0.1), network_tests.py:257-258,328-329
  simnn_iteration  the loop body SIMNN.py:276-334   (call order in SURVEY.md section 3.1)
  mmgan_iteration  the loop body network_tests.py:281-321 (call order in SURVEY.md section 3.2)
The DES bridge between G and D is replaced by explicit ``fake`` tensors (it is non-differentiable and is
entered through .detach(), SIMNN.py:299 / network_tests.py:189, so no gradient ever reaches a generator).
"""
import math

import torch


def bce_with_logits(x, y):
    """mean( max(x,0) - x*y + log1p(exp(-|x|)) )"""
    return (torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean()


class Adam:
    """Single-tensor Adam exactly as torch.optim.Adam(foreach=False, capturable=False) steps it."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.lr = lr
        self.betas = betas
        self.eps = eps
        self.state = {}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        b1, b2 = self.betas
        for p in self.params:
            if p.grad is None:
                continue  # this is what makes gen_opt.step() a no-op in both reference loops
            st = self.state.setdefault(p, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["step"] += 1
            t = st["step"]
            g = p.grad
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            step_size = self.lr / bc1
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(st["m"], denom, value=-step_size)


class StepLR:
    def __init__(self, opt, step_size, gamma=0.1):
        self.opt, self.step_size, self.gamma = opt, step_size, gamma
        self.base_lr = opt.lr
        self.epoch = 0

    def step(self):
        self.epoch += 1
        self.opt.lr = self.base_lr * self.gamma ** (self.epoch // self.step_size)


def simnn_iteration(gen, disc, gen_opt, disc_opt, real, noise, fake, elide_dead_backward=False):
    """One iteration of GAN_DES/SIMNN.py:276-334 with the bridge output given as ``fake``.

    real, fake: (B, H, W) fp32; noise: (B, 100, 1, 1).  Returns (disc_loss, gen_loss, generated) where
    ``generated`` is the (B,1,20,20) DES matrix batch the generator emitted (SIMNN.py:296).
    """
    b = real.shape[0]
    disc_opt.zero_grad()                                             # 282
    disc_real_pred = disc(real).reshape(-1)                          # 283
    real_label = torch.ones(b) * 0.9                                 # 284
    disc_real_loss = bce_with_logits(disc_real_pred, real_label)     # 289 (sigmoid output into a logits loss)
    generated = gen(noise)                                           # 296 (train mode: BN running stats move)
    disc_fake_pred = disc(fake.detach()).reshape(-1)                 # 306
    fake_label = torch.ones(b) * 0.1                                 # 308
    disc_fake_loss = bce_with_logits(disc_fake_pred, fake_label)     # 311
    disc_loss = disc_fake_loss + disc_real_loss                      # 314
    disc_loss.backward()                                             # 315
    disc_opt.step()                                                  # 316
    gen_opt.zero_grad()                                              # 322
    disc_fake_pred = disc(fake).squeeze()                            # 325
    gen_loss = bce_with_logits(disc_fake_pred, torch.ones(b))        # 326-329
    if not elide_dead_backward:
        gen_loss.backward()                                          # 330: lands in D's .grad only; wiped at 282
    gen_opt.step()                                                   # 331: every generator grad is None -> no-op
    return float(disc_loss.detach()), float(gen_loss.detach()), generated.detach()


def mmgan_iteration(mmgan, gen_opt, disc_opt, piano_roll, durations, beats, noise1, noise2, g1_in_a, g1_in_b,
                    fake_a, fake_b, count=1, elide_dead_backward=False):
    """One iteration of MMGAN_MIDI_DES/network_tests.py:281-321.

    piano_roll, durations: (B,128,T); beats: (B,50); noise1/2: (B,50); g1_in_a/b: the (B,50) tensors that
    Generator.forward draws for itself on its two calls (83-84); fake_a/b: bridge outputs (B,2,128,T) for the
    D-step and the G-step forward.  Returns (disc_loss, gen_loss, g1_out_a, g2_out_a).
    """
    b = piano_roll.shape[0]
    real = torch.ones(b)
    fake_label = torch.zeros(b)
    real_data = torch.stack([piano_roll, durations]).permute(1, 0, 2, 3)      # 290
    captured = {}

    def provider_a(g1, g2, _count):
        captured["g1"], captured["g2"] = g1, g2
        return fake_a, 0

    disc_opt.zero_grad()                                                       # 293
    mmgan.fake_provider = provider_a
    fake_output, _ = mmgan(noise1, noise2, beats, count, False, g1_input=g1_in_a)   # 294
    disc_fake_loss = bce_with_logits(fake_output.squeeze(), fake_label)        # 304
    disc_real_loss = bce_with_logits(mmgan.discriminator(real_data).squeeze(), real)  # 305
    disc_loss = disc_fake_loss + disc_real_loss
    disc_loss.backward()                                                       # 307
    disc_opt.step()                                                            # 308
    gen_opt.zero_grad()                                                        # 311
    mmgan.fake_provider = lambda g1, g2, _c: (fake_b, 0)
    fake_output, _ = mmgan(noise1, noise2, beats, count, False, g1_input=g1_in_b)   # 312 (2nd BN stat update)
    gen_loss = bce_with_logits(fake_output.squeeze(), real)                    # 313
    if not elide_dead_backward:
        gen_loss.backward()                                                    # 314
    gen_opt.step()                                                             # 315: no-op
    return float(disc_loss.detach()), float(gen_loss.detach()), captured["g1"], captured["g2"]
