"""A few eager passes of the f16 head at 640x480 batch 8 with decoder.fused_head_masks = MSM_FUSED (0 / 1) -- under rocprofv3 --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
model.set_precision(os.environ.get("MSM_PRECISION", "f16"))
model.sem_seg_head.predictor.fused_head_masks = bool(int(os.environ.get("MSM_FUSED", "0")))
model.sem_seg_head.predictor.weight_prefetch = bool(int(os.environ.get("MSM_PREFETCH", "1")))
for _ in range(10):
    model.inference(feats, (480, 640))
torch.cuda.synchronize()
