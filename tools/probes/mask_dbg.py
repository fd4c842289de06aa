import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import _lib, ops
DEV = "cuda"
torch.manual_seed(0)
for (B, Q, C, H, W, pool) in [(1, 100, 256, 120, 160, 2), (1, 100, 64, 120, 160, 2), (1, 100, 256, 120, 160, 0), (1, 100, 32, 16, 32, 0)]:
    e = torch.randn(B, Q, C, device=DEV) * 0.3
    f = torch.randn(B, C, H, W, device=DEV)
    tgt = None if pool == 0 else (H // pool, W // pool)
    outs = {}
    for nc in (1, 2):
        _lib.set_option("MASK_NC", nc)
        outs[nc] = ops.mask_logits(e, f, want_mask=True, target_size=tgt)
    ref = torch.einsum("bqc,bchw->bqhw", e.double(), f.double()).float()
    for nc in (1, 2):
        bad = (outs[nc][0] - ref).abs() > 1e-3
        print(f"B{B} Q{Q} C{C} {H}x{W} pool{pool} nc={nc}: bad {int(bad.sum())}")
        if bad.any():
            idx = bad.nonzero()
            qs, ys, xs = idx[:, 1], idx[:, 2], idx[:, 3]
            print("   q%16 hist", torch.bincount(qs % 16, minlength=16).tolist())
            print("   q//16 hist", torch.bincount(qs // 16, minlength=7).tolist())
            print("   x%32 hist", torch.bincount(xs % 32, minlength=32).tolist())
            print("   y%2 hist", torch.bincount(ys % 2, minlength=2).tolist(), "rows", torch.unique(ys).tolist()[:20], "xtile", torch.unique(xs // 32).tolist())
            print("   sample got/ref", outs[nc][0][bad][:5].tolist(), ref[bad][:5].tolist())
    if tgt is not None:
        print("   attn equal nc1 vs nc2:", torch.equal(outs[1][1], outs[2][1]), "row_any", torch.equal(outs[1][2], outs[2][2]))
