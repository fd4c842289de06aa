"""configs[3] (two-stage refinement over 16 frames): wall time per stage of two_stage.test_batch_crop_nolabel, host-synchronised after
each stage, per precision plan -- where the batch's milliseconds go (first stage / label image + depth filter / ROI transfer / crops /
second stage / match + paste)."""
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
marks = []


def mark(tag):
    torch.cuda.synchronize()
    marks.append((tag, time.perf_counter()))


class Pred(Network_RGBD):
    def batch_tensors(self, samples):
        imgs = torch.stack([x["image"] for x in samples])
        deps = torch.stack([x["depth"] for x in samples])
        mark(f"stack {len(samples)}")
        with torch.no_grad():
            f = self.model.backbone(imgs, deps)
            mark("backbone")
            sc, cl, mk, _, _ = self.model.inference(f, tuple(int(v) for v in imgs.shape[-2:]))
        mark(f"head {len(samples)}")
        return sc, cl, mk


p = Pred(rgbd)
gen = torch.Generator().manual_seed(3)
samples = [{"image_color": torch.rand(3, H, W, generator=gen).to(dev), "depth": torch.rand(3, H, W, generator=gen).to(dev)} for _ in range(16)]
for mode in ("f32", "f16"):
    rgbd.set_precision(mode)
    for _ in range(3):
        ts.test_batch_crop_nolabel(samples, p, p, confident_score=0.0, topk=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ts.test_batch_crop_nolabel(samples, p, p, confident_score=0.0, topk=False)
    torch.cuda.synchronize()
    print(f"{mode}: unsynchronised {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per batch of 16")
    orig = {}
    for name in ("_label_image_batched", "filter_labels_depth", "label_stats", "roi_table", "_crop_resize_batched", "match_label_crop_batched"):
        orig[name] = getattr(ts, name)

        def wrap(fn, name=name):
            def w(*a, **k):
                r = fn(*a, **k)
                mark(name)
                return r
            return w
        setattr(ts, name, wrap(orig[name]))
    marks.clear()
    mark("start")
    out = ts.test_batch_crop_nolabel(samples, p, p, confident_score=0.0, topk=False)
    mark("end")
    for name, fn in orig.items():
        setattr(ts, name, fn)
    print(f"{mode}: {len(out[2])} crops")
    for (a, ta), (b, tb) in zip(marks[:-1], marks[1:]):
        print(f"    {b:32s} {1e3 * (tb - ta):7.3f} ms")
    print(f"    total (synchronised)             {1e3 * (marks[-1][1] - marks[0][1]):7.3f} ms")
