"""configs[4] hot path under HIP-graph replay: batch 1 and 4, f32 and bf16 (what bench.py reports; tuning aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda")
model = bench.build_model(dev, num_queries=300, dec_layers=20)
for B in (1, 4):
    feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(B, 960, 1280, seed=9).items()}
    for mode in ("f32", "bf16"):
        model.set_precision(mode)
        g = model.graphed()
        for _ in range(3):
            g(feats, (960, 1280))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g(feats, (960, 1280))
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20
        print(f"B={B} {mode}: {1e3 * t:.3f} ms per batch, {B / t:.1f} images/s")
        del g
