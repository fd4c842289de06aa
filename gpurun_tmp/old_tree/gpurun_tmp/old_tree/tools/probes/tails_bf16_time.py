"""The bf16 decoder tails at B = 8, Q = 100 under HIP-graph timing (tuning aid; tools/probes/tails_bf16_parts.sh switches parts off)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402
from microbench import timeit_graph  # noqa: E402

DEV = "cuda:0"
if len(sys.argv) > 2 and sys.argv[1] == "--lp-f16":          # MSM_OPT_LP_F16: 1 = fp16 weights + hi / lo fp16 activations, 2 = one fp16 activation term
    from unseenobjectswithmeanshift_amd import _lib
    _lib.set_option("LP_F16", int(sys.argv[2]))
    print(f"LP_F16 = {sys.argv[2]}")
B, Q, E, Fh = 8, 100, 256, 2048
r = lambda *s: torch.randn(*s, device=DEV)
o, res, qpos = r(B, Q, E), r(B, Q, E), r(Q, E)
for name, pk in (("bf16", lambda n, k: ops.dec_pack_weight_bf16(r(n, k) * k ** -0.5)), ("f32", lambda n, k: ops.dec_pack_weight(r(n, k) * k ** -0.5))):
    wo, w_in, w1, w2, wq = pk(E, E), pk(3 * E, E), pk(Fh, E), pk(E, Fh), pk(E, E)
    mlp = [(pk(E, E), r(E)) for _ in range(3)]
    v = lambda: r(E)
    bo, g, b, b_in, b1, b2 = v(), v(), v(), r(3 * E), r(Fh), v()
    t1 = timeit_graph(lambda: ops.dec_post_cross(o, res, qpos, wo, bo, g, b, w_in, b_in))
    t2 = timeit_graph(lambda: ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2))
    x, parts = ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2)
    t3 = timeit_graph(lambda: ops.dec_heads(x, g, b, mlp, parts=parts, bias=b2, ln_g=g, ln_b=b, l2norm=True, wq=wq, bq=bo, query_pos=qpos))
    print(f"{name}: post_cross {t1:5.1f} us, post_self {t2:5.1f} us ({parts.shape[0]} parts), heads {t3:5.1f} us")
