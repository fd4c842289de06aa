"""Input projections at B = 8, 640x480: fp32 MFMA kernels against the hi + lo bf16 form (tuning aid).  The inputs rotate through
enough copies (> 256 MB of MALL) that every launch streams from HBM, as in a real pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402

B = 8
dev = "cuda"
shapes = ((2048, 15, 20), (1024, 30, 40), (512, 60, 80))
NC = 4
xs = [[torch.randn(B, c, h, w, device=dev) for c, h, w in shapes] for _ in range(NC)]
wsf = [torch.randn(64, c, device=dev) * c ** -0.5 for c, _, _ in shapes]
bs = [torch.randn(64, device=dev) for _ in shapes]
S = sum(h * w for _, h, w in shapes)
out = torch.empty(B, S, 64, device=dev)
st = torch.zeros(3, B, 64, 2, device=dev, dtype=torch.float64)
x2 = [torch.randn(B, 256, 120, 160, device=dev) for _ in range(NC)]
w2 = torch.randn(64, 256, device=dev) / 16


def t(fn, reps=20):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for lp in (False, True):
    pack = ops.pack_conv_in_weight_lp if lp else ops.pack_conv_in_weight
    ws = [pack(w) for w in wsf]
    tm = t(lambda i: ops.conv1x1_in_multi(xs[i % NC], ws, bs, out, st, stats_cleared=True, lp=lp))
    wl = pack(w2)
    ts = t(lambda i: ops.conv1x1_in(x2[i % NC], wl, None, lp=lp))
    per = [t(lambda i: ops.conv1x1_in(xs[i % NC][l], ws[l], bs[l], lp=lp)) for l in range(3)]
    print(f"lp={lp}: multi {tm:.1f} us (137 MB: {137.6e6 / tm / 1e6:.2f} TB/s), lateral {ts:.1f} us ({157.3e6 / ts / 1e6:.2f} TB/s), levels alone {[round(p, 1) for p in per]}")
