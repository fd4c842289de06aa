"""The fp16 decoder tails at large row counts (the second stage of configs[3]: 173 crops x 100 queries): 16-row tiles against the 32-row
tiles of round 6 (MSM_OPT_DEC_TILE32), event-timed per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
E, Fh = 256, 2048
g_ = torch.Generator().manual_seed(1)
r = lambda *s, k=1.0: (torch.randn(*s, generator=g_) * k).to(dev)
pk = ops.dec_pack_weight_f16
wo, bo, g, b = pk(r(E, E, k=E ** -0.5)), r(E, k=0.1), 1 + r(E, k=0.1), r(E, k=0.1)
w_in, b_in = pk(r(3 * E, E, k=E ** -0.5)), r(3 * E, k=0.1)
w1, b1, w2, b2 = pk(r(Fh, E, k=E ** -0.5)), r(Fh, k=0.1), pk(r(E, Fh, k=Fh ** -0.5)), r(E, k=0.1)
g1, be1, g2, be2 = 1 + r(E, k=0.1), r(E, k=0.1), 1 + r(E, k=0.1), r(E, k=0.1)
mlp = [(pk(r(E, E, k=E ** -0.5)), r(E, k=0.1)) for _ in range(3)]
wq, bq = pk(r(E, E, k=E ** -0.5)), r(E, k=0.1)


def ev(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for B in (48, 100, 173, 192):
    Q = 100
    o, res, qpos = r(B, Q, E), r(B, Q, E), r(Q, E)
    x2, parts = ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2)
    line = [f"rows {B * Q:6d}"]
    for t32 in (0, 1):
        with _lib.option("DEC_TILE32", t32):
            t_c = ev(lambda: ops.dec_post_cross(o, res, qpos, wo, bo, g, b, w_in, b_in))
            t_s = ev(lambda: ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2))
            t_h = ev(lambda: ops.dec_heads(x2, g2, be2, mlp, parts=parts, bias=b2, ln_g=g1, ln_b=be1, l2norm=True, wq=wq, bq=bq, query_pos=qpos, zero_row_any=True))
        line.append(f"{('16', '32')[t32]}-row tiles: post_cross {t_c:6.1f} post_self {t_s:6.1f} heads {t_h:6.1f} us")
    print(" | ".join(line))
