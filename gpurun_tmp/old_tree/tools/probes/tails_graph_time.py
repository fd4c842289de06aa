"""The three fp16 / bf16 decoder tails at 800 rows (B = 8, Q = 100), each as 20 back-to-back launches replayed from a HIP graph
(kernel time without host launch gaps): for A/B runs of two builds on the same box."""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("MSM_TREE", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
E, Fh, B, Q = 256, 2048, 8, 100
g_ = torch.Generator().manual_seed(1)
r = lambda *s, k=1.0: (torch.randn(*s, generator=g_) * k).to(dev)


def graph_ms(fn, n=20, reps=50):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps / n


for name, pk in (("f16", ops.dec_pack_weight_f16), ("bf16", ops.dec_pack_weight_bf16)):
    wo, bo, g, b = pk(r(E, E, k=E ** -0.5)), r(E, k=0.1), 1 + r(E, k=0.1), r(E, k=0.1)
    w_in, b_in = pk(r(3 * E, E, k=E ** -0.5)), r(3 * E, k=0.1)
    w1, b1, w2, b2 = pk(r(Fh, E, k=E ** -0.5)), r(Fh, k=0.1), pk(r(E, Fh, k=Fh ** -0.5)), r(E, k=0.1)
    g1, be1, g2, be2 = 1 + r(E, k=0.1), r(E, k=0.1), 1 + r(E, k=0.1), r(E, k=0.1)
    mlp = [(pk(r(E, E, k=E ** -0.5)), r(E, k=0.1)) for _ in range(3)]
    wq, bq = pk(r(E, E, k=E ** -0.5)), r(E, k=0.1)
    o, res, qpos = r(B, Q, E), r(B, Q, E), r(Q, E)
    x2, parts = ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2)
    t_c = graph_ms(lambda: ops.dec_post_cross(o, res, qpos, wo, bo, g, b, w_in, b_in))
    t_s = graph_ms(lambda: ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2))
    t_h = graph_ms(lambda: ops.dec_heads(x2, g2, be2, mlp, parts=parts, bias=b2, ln_g=g1, ln_b=be1, l2norm=True, wq=wq, bq=bq, query_pos=qpos, zero_row_any=True))
    print(f"{name}: post_cross {t_c:6.2f} post_self {t_s:6.2f} heads {t_h:6.2f} us per launch (20 back to back in a graph)")
