#!/usr/bin/env python3
"""What does HBM give for write-heavy streams?  torch's own fill / copy / convert kernels at conv1-forward's sizes
(33.5 MB read, 84 MB written per B = 256 launch) -- an upper-bound indication for kernels whose traffic is mostly stores."""
import torch
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
dev = "cuda"
MB = 1 << 20
w = torch.empty(84 * MB, dtype=torch.uint8, device=dev)
w2 = torch.empty(84 * MB, dtype=torch.uint8, device=dev)
r = torch.randn(int(33.5 * MB / 4), device=dev)
o = torch.empty(r.numel() * 4, dtype=torch.bfloat16, device=dev)          # 2 x the bytes of r written, like p1
big = torch.empty(252 * MB // 2, dtype=torch.uint8, device=dev)
big2 = torch.empty_like(big)
us = t(lambda: w.fill_(1)); print(f"fill 84 MB                 {us:6.1f} us  {84 * MB / us / 1e6:5.2f} TB/s written")
us = t(lambda: w2.copy_(w)); print(f"copy 84 MB -> 84 MB        {us:6.1f} us  {168 * MB / us / 1e6:5.2f} TB/s total")
us = t(lambda: big2.copy_(big)); print(f"copy 126 MB -> 126 MB      {us:6.1f} us  {252 * MB / us / 1e6:5.2f} TB/s total")
us = t(lambda: torch.sum(big.view(torch.int32))); print(f"read 126 MB (sum)          {us:6.1f} us  {126 * MB / us / 1e6:5.2f} TB/s read")
ov = o.view(4, -1)
us = t(lambda: ov.copy_(r.unsqueeze(0).expand(4, -1))); print(f"read 33.5 MB, write 67 MB   {us:6.1f} us  {100.5 * MB / us / 1e6:5.2f} TB/s total")
