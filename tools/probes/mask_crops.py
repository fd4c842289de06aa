"""Mask step at the second-stage crop geometry of configs[3] (B = 171, 56 x 56 mask features, folded C = 64): tuning aid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import _lib, ops  # noqa: E402
from microbench import timeit_graph  # noqa: E402

dev = "cuda"
for B, H, W in ((171, 56, 56), (171, 64, 64), (64, 56, 56), (8, 120, 160)):
    wide = torch.randn(B, 100, 256, device=dev) * 0.3
    f = torch.randn(B, 64, H, W, device=dev)
    for pool in (8, 4, 2, 0):
        tgt = None if pool == 0 else (H // pool, W // pool)
        line = f"B={B} {H}x{W} pool {pool}:"
        for nc in (-1, 1, 2):
            with _lib.option("MASK_NC", nc):
                t = timeit_graph(lambda: ops.mask_logits(wide[..., :64], f, want_mask=pool == 0, target_size=tgt, qbias=wide[..., 64]), reps=10)
            line += f"  nc={nc}: {t:7.1f} us"
        fl = 2.0 * 100 * 64 * H * W * B
        print(line + f"   ({fl / 1e9:.2f} GFLOP)", flush=True)
