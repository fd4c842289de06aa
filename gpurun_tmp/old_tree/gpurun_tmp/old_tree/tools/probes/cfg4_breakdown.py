"""configs[4] hot path (1280x960, 300 queries, 20 decoder layers): per-entry-point launch times, batch 1 and 4, f32 and bf16 (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import _lib, synthetic as syn  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda")
model = bench.build_model(dev, num_queries=300, dec_layers=20)
for B in (1, 4):
    feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(B, 960, 1280, seed=9).items()}
    for mode in ("f16",):
        model.set_precision(mode)
        for _ in range(2):
            model.inference(feats, (960, 1280))
        with _lib.CallTimer() as ct:
            model.inference(feats, (960, 1280))
            torch.cuda.synchronize()
        d = ct.durations()
        tot = sum(sum(v) for v in d.values())
        print(f"B={B} {mode}: sum of launches {tot:.3f} ms")
        for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
            print(f"    {k:40s} x{len(v):3d}  {sum(v):7.3f} ms  ({1e3 * sum(v) / len(v):6.1f} us each)")
