#!/usr/bin/env python
"""HBM bandwidth calibration on the GPU box: fill (write only), copy (read + write), sum (read only) of a 1 GiB buffer
through torch's own kernels.  Gives the practical ceilings the write-bound kernels (mask upsample, K/V projection,
mask_features) are compared against in DESIGN.md."""
import torch

x = torch.empty(256 << 20, device="cuda")          # 1 GiB fp32
y = torch.empty_like(x)


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


gib = x.numel() * 4
print(f"fill  (write):      {gib / t(lambda: x.fill_(1.0)) / 1e12:5.2f} TB/s")
print(f"copy  (read+write): {2 * gib / t(lambda: y.copy_(x)) / 1e12:5.2f} TB/s")
print(f"sum   (read):       {gib / t(lambda: x.sum()) / 1e12:5.2f} TB/s")
