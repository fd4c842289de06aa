"""configs[3] timing by crop batch size, with a per-phase breakdown (tuning aid): python tools/probes/twostage_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn, two_stage as ts  # noqa: E402
from unseenobjectswithmeanshift_amd.meta_arch import MeanShiftMaskFormer, Network_RGBD  # noqa: E402

dev = torch.device("cuda", 0)
H, W = 480, 640
model = bench.build_model(dev)
bb = syn.StandInBackbone().to(dev).eval()
rgbd = MeanShiftMaskFormer(backbone=bb, sem_seg_head=model.sem_seg_head, num_queries=100)
T = {}


class Pred(Network_RGBD):
    def batch_tensors(self, samples):
        imgs = torch.stack([x["image"] for x in samples])
        deps = torch.stack([x["depth"] for x in samples])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            sc, cl, mk, _, _ = self.model.inference(self.model.backbone(imgs, deps), tuple(int(v) for v in imgs.shape[-2:]))
        torch.cuda.synchronize(); T.setdefault(len(samples), []).append(time.perf_counter() - t0)
        return sc, cl, mk


p = Pred(rgbd)
gen = torch.Generator().manual_seed(3)
samples = [{"image_color": torch.rand(3, H, W, generator=gen).to(dev), "depth": torch.rand(3, H, W, generator=gen).to(dev)} for _ in range(16)]
for cb in (32, 64, 96, 128, 192):
    for _ in range(2):
        ts.test_batch_crop_nolabel(samples, p, p, confident_score=0.0, topk=False, crop_batch=cb)
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        ts.test_batch_crop_nolabel(samples, p, p, confident_score=0.0, topk=False, crop_batch=cb)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
    calls = {k: round(1e3 * sum(v) / len(v), 2) for k, v in sorted(T.items())}
    print(f"crop_batch {cb:4d}: {1e3 * t:6.2f} ms per batch of 16 (incl. the probe's syncs); predictor calls by batch size (ms): {calls}", flush=True)
