"""torch.optim-compatible Adam whose update runs in the fused HIP kernel (gdm_adam_step).

Drop-in for the reference's ``torch.optim.Adam(params, lr=..., betas=...)`` (GAN_DES/SIMNN.py:258-259,
MMGAN_MIDI_DES/network_tests.py:253-254): same constructor, ``param_groups`` (so ``StepLR`` works unchanged),
``zero_grad``/``step``, parameters without a gradient are skipped (which is what makes the reference's
``gen_opt.step()`` a no-op).  The fused trainers in ``train.py`` do not use this class: they step one flat buffer.
"""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise NotImplementedError("weight_decay is not used on the reference's path")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise ops.GdmError("fused Adam needs contiguous parameters and gradients")
                ops.adam_step(p.data.view(-1), p.grad.view(-1), st["exp_avg"].view(-1), st["exp_avg_sq"].view(-1),
                              st["step"], group["lr"], b1, b2, group["eps"])
        return loss
