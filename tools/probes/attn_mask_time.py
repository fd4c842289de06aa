"""Attention masks at key resolution: the pooling launch and the three key counts (tuning aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops
from microbench import timeit_graph
DEV = "cuda:0"
B, Q = 8, 100
act = torch.randn(B, 64, 120, 160, device=DEV)
wide = torch.randn(B, Q, 256, device=DEV) * 0.3
sizes = [(15, 20), (30, 40), (60, 80)]
print(f"pool_mask_taps (3 levels): {timeit_graph(lambda: ops.pool_mask_taps(act, sizes)):.1f} us")
pooled = ops.pool_mask_taps(act, sizes)
ra = torch.zeros(B, Q, device=DEV, dtype=torch.int32)
for s, p in zip(sizes, pooled):
    t = timeit_graph(lambda: ops.attn_mask_pooled(wide[..., :64], p, qbias=wide[..., 64], row_any=ra))
    t2 = timeit_graph(lambda: ops.mask_logits(wide[..., :64], act, want_mask=False, target_size=s, qbias=wide[..., 64], row_any=ra))
    print(f"attn_mask_pooled {s}: {t:.1f} us   (full-resolution kernel: {t2:.1f} us)")
