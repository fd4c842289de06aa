"""The fused K/V + attention kernel at the UCN size (batch 2, 480 x 640 keys, 100 queries, masked) under HIP-graph timing, next to the
unfused pair it replaces (tuning aid; tools/probes/fkv_parts.sh switches parts off)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402
from microbench import timeit_graph  # noqa: E402

DEV = "cuda"
B, Lq, H, W, E, Hh = 2, 100, 480, 640, 256, 8
S = H * W
g = torch.Generator(device=DEV).manual_seed(0)
r = lambda *s: torch.randn(*s, device=DEV, generator=g)
q = r(B, Lq, E)
x = torch.nn.functional.normalize(r(B, 64, H, W), dim=1)
w = r(2 * E, 64) * 0.3
rowcol = r(H + W, 2 * E) * 0.3
masked = (torch.rand(B, Lq, S, device=DEV, generator=g) < 0.6).to(torch.uint8)
row_any = torch.ones(B, Lq, dtype=torch.int32, device=DEV)
xh = ops.tokens_f16(x)
wp = ops.attn_pack_kv_weights(w, Hh)
cvt = rowcol[H:, E:].t().contiguous()
bits = ops.attn_pack_mask_bits(masked)                  # (what the mask producers of the 16-bit plans hand over)
for kf in (False, True):
    t = timeit_graph(lambda: ops.hypersphere_attention_fused_kv(q, xh, wp, rowcol, cvt, (H, W), Hh, masked=masked, row_any=row_any, keys_f16=kf), reps=5, iters=5)
    tb = timeit_graph(lambda: ops.hypersphere_attention_fused_kv(q, xh, wp, rowcol, cvt, (H, W), Hh, masked=bits, row_any=row_any, keys_f16=kf), reps=5, iters=5)
    print(f"fused K/V attention, keys_f16={kf}: {t:8.1f} us with the mask packed per call, {tb:8.1f} us on pre-packed bits", flush=True)
if "fused-only" not in sys.argv:
    for kf in (False, True):
        kv = ops.kv_project_multi([x], [w], [rowcol], out_dtype=torch.bfloat16, cmat_widths=[W], keys_f16=kf)[0]
        t1 = timeit_graph(lambda: ops.kv_project_multi([x], [w], [rowcol], out_dtype=torch.bfloat16, cmat_widths=[W], keys_f16=kf), reps=5, iters=5)
        t2 = timeit_graph(lambda: ops.hypersphere_attention(q, kv[..., :E], kv[..., E:], Hh, masked=masked, row_any=row_any, low_precision=True, keys_f16=kf), reps=5, iters=5)
        print(f"unfused, keys_f16={kf}: projection {t1:8.1f} us + attention {t2:8.1f} us = {t1 + t2:8.1f} us", flush=True)
    t = timeit_graph(lambda: ops.tokens_f16(x), reps=5, iters=5)
    print(f"tokens_f16 (once per forward): {t:8.1f} us")
