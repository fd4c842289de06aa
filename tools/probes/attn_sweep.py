"""Attention kernel choices at the decoder's key counts (tuning aid): key splits of the split-K kernel, query-split kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops, _lib
from microbench import timeit_graph
DEV = "cuda:0"
B, E = 8, 256
for S in (4800, 1200, 300):
    q, k = torch.randn(B, 100, E, device=DEV), torch.randn(B, S, 2 * E, device=DEV)
    m = (torch.rand(B, 100, S, device=DEV) < 0.5).to(torch.uint8)
    ra = torch.ones(B, 100, device=DEV, dtype=torch.int32)
    run = lambda: ops.hypersphere_attention(q, k[..., :E], k[..., E:], 8, masked=m, row_any=ra)
    print(f"S={S}: default {timeit_graph(run):.1f} us", flush=True)
    with _lib.option("ATTN_KERNEL", 3):
        for tgt in (256, 384, 512, 640, 768, 1024):
            with _lib.option("ATTN_TARGET", tgt):
                print(f"   split-K kernel, target {tgt}: {timeit_graph(run):.1f} us", flush=True)
    with _lib.option("ATTN_QK_MAX", 8192):
        for cfg in (0, 1):
            with _lib.option("ATTN_QKCFG", cfg):
                print(f"   query-split kernel, {2 - cfg} query blocks per workgroup: {timeit_graph(run):.1f} us", flush=True)
