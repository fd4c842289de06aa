"""The UCN RGB-D towers (two dilated ResNet34-8s, batch 2 at 480x640): elementwise glue as torch ops against one HIP launch each (tuning aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402
from unseenobjectswithmeanshift_amd.ucn_backbone import UCNBackbone  # noqa: E402

bb = UCNBackbone(num_units=64, in_channels=3, use_depth=True).to("cuda").eval()
bb.load_state_dict(syn.ucn_backbone_state_dict(syn.ucn_backbone_param_shapes(), salt=6), strict=True)
g = torch.Generator().manual_seed(5)
img, dep = torch.randn(2, 3, 480, 640, generator=g).cuda(), torch.rand(2, 3, 480, 640, generator=g).cuda()


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


ref = None
for dt in ("bf16", "f16", "f32"):
    bb.backbone_dtype = dt
    for fused, par in ((False, False), (True, False), (True, True)):
        bb.fused_epilogues, bb.parallel_towers = fused, par
        out = bb(img, None, dep)
        if not fused:
            ref = out
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            bb(img, None, dep)
            s.synchronize()
            with torch.cuda.graph(g, stream=s):
                bb(img, None, dep)
        torch.cuda.synchronize()
        print(f"{dt} fused_epilogues={fused} parallel_towers={par}: eager {timed(lambda: bb(img, None, dep)):.3f} ms, graph {timed(g.replay, 20):.3f} ms"
              + ("" if not fused else f"; max |d| against the torch ops {float((out - ref).abs().max()):.2e}"), flush=True)
