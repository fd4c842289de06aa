"""What a kernel boundary costs inside a replayed HIP graph on this part: chains of N dependent trivial launches (a 1-element add; a
256-row LayerNorm of the library = one short real kernel), time per node (tuning aid: the 16-bit head pass is 85 dependent launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402

dev = torch.device("cuda")
x1 = torch.zeros(1, device=dev)
x = torch.randn(800, 256, device=dev)
g_, b_ = torch.ones(256, device=dev), torch.zeros(256, device=dev)


def chain(fn, n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3


for name, fn in (("x += 1 on one element", lambda: x1.add_(1.0)), ("LayerNorm of 800 x 256 (library kernel)", lambda: ops.layernorm(x, g_, b_))):
    t10, t210 = chain(fn, 10), chain(fn, 210)
    print(f"{name}: {(t210 - t10) / 200:.2f} us per dependent node in a replayed graph (10 nodes {t10:.1f} us, 210 nodes {t210:.1f} us)", flush=True)
