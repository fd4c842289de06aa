"""configs[3] (two-stage refinement over 16 frames, bench.py's harness) a few times -- run under rocprofv3 --kernel-trace --stats
(tools/probes/stats_table.py prints the per-kernel table)."""
import os
import sys

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


class Pred(Network_RGBD):
    def batch_tensors(self, samples):
        imgs = torch.stack([x["image"] for x in samples])
        deps = torch.stack([x["depth"] for x in samples])
        with torch.no_grad():
            sc, cl, mk, _, _ = self.model.inference(self.model.backbone(imgs, deps), tuple(int(v) for v in imgs.shape[-2:]))
        return sc, cl, mk


p = Pred(rgbd)
rgbd.set_precision(os.environ.get("MSM_PRECISION", "f32"))
gen = torch.Generator().manual_seed(3)
samples = [{"image_color": torch.rand(3, H, W, generator=gen).to(dev), "depth": torch.rand(3, H, W, generator=gen).to(dev)} for _ in range(16)]
for _ in range(6):
    ts.test_batch_crop_nolabel(samples, p, p, confident_score=0.0, topk=False)
torch.cuda.synchronize()
