"""ResNet-50 backbone, batch 8 at 480x640: the elementwise glue as torch ops against one HIP launch each (fused_epilogues), fp32 and bf16,
eager and from a HIP graph (tuning aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd.resnet_backbone import ResNet50Backbone  # noqa: E402

bb = ResNet50Backbone().to("cuda").eval()
images = torch.randn(8, 3, 480, 640, device="cuda")


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for dt in ("bf16", "f16", "f32"):
    bb.backbone_dtype = dt
    for fused in (False, True):
        bb.fused_epilogues = fused
        te = timed(lambda: bb(images))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            bb(images)
            s.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = bb(images)
        torch.cuda.synchronize()
        tg = timed(g.replay, 20)
        print(f"{dt} fused_epilogues={fused}: eager {te:.3f} ms, graph {tg:.3f} ms", flush=True)
