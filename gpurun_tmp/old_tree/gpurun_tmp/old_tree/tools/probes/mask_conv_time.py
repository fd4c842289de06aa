"""msm_mask_conv3x3_folded at the UCN shapes (batch 2 of 480x640, Q = 100 bits / K = 20 logits): time per launch (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402

dev = torch.device("cuda")
B, H, W = 2, 480, 640
g = torch.Generator().manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(B, 64, H, W, generator=g), dim=1).to(dev)
xh = ops.tokens_f16(x)
wf = ops.mask_conv_fold_weight((torch.randn(256, 64, 3, 3, generator=g) / 24).to(dev), (torch.randn(256, generator=g) * 0.1).to(dev))


def timed(fn, n=60):
    for _ in range(40):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for Q, bits in ((100, True), (20, False), (100, False), (112, True), (16, True), (100, True)):
    e = torch.randn(B, Q, 256, generator=g).to(dev)
    Fq = ops.gemm(e, wf)
    ra = torch.zeros(B, Q, device=dev, dtype=torch.int32)
    t = timed(lambda: ops.mask_conv3x3_folded(xh, Fq, (H, W), bits=bits, row_any=ra if bits else None))
    fl = 2.0 * B * (16 * ((Q + 15) // 16)) * 576 * H * W
    print(f"Q={Q:3d} {'bits  ' if bits else 'logits'}: {t:7.1f} us per launch = {fl / t / 1e6:6.1f} TFLOP/s executed ({fl / t / 1e6 / 2516.6:.3f} of the f16 MFMA peak)", flush=True)
