"""One pass of the head at 640x480, batch 8, in the 16-bit plans with and without decoder.fused_head_masks (round 6): graph-replayed
step time, one batch in flight."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
for mode, lp in (("f16", False), ("bf16", False), ("f32", False)):
    model.set_precision(mode)
    for fused in (False, True, "always", False, True, "always"):
        model.sem_seg_head.predictor.weight_prefetch = fused
        g = model.graphed()
        for _ in range(5):
            g(feats, (480, 640))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g(feats, (480, 640))
        torch.cuda.synchronize()
        print(f"{mode} weight_prefetch={fused}: {1e3 * (time.perf_counter() - t0) / 200:.4f} ms per batch of 8")
        del g
