"""Per-entry-point time of one second-stage call of configs[3] (171 crops of 224x224) and of the first stage (16 frames): tuning aid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import _lib, synthetic as syn  # noqa: E402
from unseenobjectswithmeanshift_amd.meta_arch import MeanShiftMaskFormer  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
bb = syn.StandInBackbone().to(dev).eval()
rgbd = MeanShiftMaskFormer(backbone=bb, sem_seg_head=model.sem_seg_head, num_queries=100)
for B, hw in ((171, 224), (16, None)):
    h, w = (hw, hw) if hw else (480, 640)
    imgs, deps = torch.rand(B, 3, h, w, device=dev), torch.rand(B, 3, h, w, device=dev)
    with torch.no_grad():
        for _ in range(2):
            rgbd.inference(rgbd.backbone(imgs, deps), (h, w))
        torch.cuda.synchronize()
        with _lib.CallTimer() as ct:
            feats = rgbd.backbone(imgs, deps)
            rgbd.inference(feats, (h, w))
            torch.cuda.synchronize()
    d = ct.durations()
    tot = sum(sum(v) for v in d.values())
    print(f"B={B} {h}x{w}: library launches {tot:.2f} ms")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print(f"   {k:36s} {len(v):3d} launches {sum(v):7.3f} ms")
