"""Is the gap between a tail's launch inside a pass (14 us) and back to back (9.7 us) the weights' residency?  post_self / heads / post_cross
(f16, 800 rows) timed by HIP events after the caches were flushed by a 512-MB copy, with and without ops.l2_prefetch of the launch's
weights in between (sequential, same stream: the question is the residency, not the overlap)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
E, Fh, B, Q = 256, 2048, 8, 100
g_ = torch.Generator().manual_seed(1)
r = lambda *s, k=1.0: (torch.randn(*s, generator=g_) * k).to(dev)
pk = ops.dec_pack_weight_f16
wo, bo, g, b = pk(r(E, E, k=E ** -0.5)), r(E, k=0.1), 1 + r(E, k=0.1), r(E, k=0.1)
w_in, b_in = pk(r(3 * E, E, k=E ** -0.5)), r(3 * E, k=0.1)
w1, b1, w2, b2 = pk(r(Fh, E, k=E ** -0.5)), r(Fh, k=0.1), pk(r(E, Fh, k=Fh ** -0.5)), r(E, k=0.1)
g1, be1, g2, be2 = 1 + r(E, k=0.1), r(E, k=0.1), 1 + r(E, k=0.1), r(E, k=0.1)
mlp = [(pk(r(E, E, k=E ** -0.5)), r(E, k=0.1)) for _ in range(3)]
wq, bq = pk(r(E, E, k=E ** -0.5)), r(E, k=0.1)
wo2, w_in2 = pk(r(E, E, k=E ** -0.5)), pk(r(3 * E, E, k=E ** -0.5))      # (the launch in front streams weights of its own)
o, res, qpos = r(B, Q, E), r(B, Q, E), r(Q, E)
x2, parts = ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2)
big_a, big_b = torch.empty(128 << 20, device=dev), torch.empty(128 << 20, device=dev)

cases = {
    "post_cross": (lambda: ops.dec_post_cross(o, res, qpos, wo, bo, g, b, w_in, b_in), [wo, w_in]),
    "post_self": (lambda: ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2), [wo, w1, w2]),
    "heads": (lambda: ops.dec_heads(x2, g2, be2, mlp, parts=parts, bias=b2, ln_g=g1, ln_b=be1, l2norm=True, wq=wq, bq=bq, query_pos=qpos, zero_row_any=True),
              [w for w, _ in mlp] + [wq]),
}
for name, (fn, ws) in cases.items():
    for mode in ("cold", "weights prefetched", "weights prefetched by a row of a post_cross launch in front", "cold, a post_cross launch in front"):
        ts = []
        for it in range(12):
            if not mode.startswith("weights +"):
                big_b.copy_(big_a)                       # 1 GB of traffic: L2 and the Infinity Cache turn over
            if mode == "weights prefetched":
                ops.l2_prefetch(ws)
            elif "post_cross launch" in mode:
                if "row" in mode:
                    ops.dec_set_prefetch(ws)
                ops.dec_post_cross(o, res, qpos, wo2, bo, g, b, w_in2, b_in)
            else:
                ops.l2_prefetch([bo])                    # (a launch in the same place either way)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            if it >= 2:
                ts.append(1e3 * e0.elapsed_time(e1))
        ts.sort()
        print(f"{name:10s} {mode:60s}: median {ts[len(ts) // 2]:6.2f} us (min {ts[0]:6.2f})")
