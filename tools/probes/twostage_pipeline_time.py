"""configs[3] through two_stage.BatchedTwoStage (both stages from HIP graphs, two batches in flight) per precision plan -- what bench.py's
c3 entry times."""
import os
import sys
import time

import torch

if os.environ.get("MSM_TREE"):                         # A/B of two builds on one box
    sys.path.insert(0, os.path.abspath(os.environ["MSM_TREE"]))
sys.path.append(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn, two_stage as ts  # noqa: E402
from unseenobjectswithmeanshift_amd.meta_arch import MeanShiftMaskFormer  # noqa: E402

dev = torch.device("cuda", 0)
H, W = 480, 640
model = bench.build_model(dev)
rgbd = MeanShiftMaskFormer(backbone=syn.StandInBackbone().to(dev).eval(), sem_seg_head=model.sem_seg_head, num_queries=100)
gen = torch.Generator().manual_seed(3)
samples = [{"image_color": torch.rand(3, H, W, generator=gen).to(dev), "depth": torch.rand(3, H, W, generator=gen).to(dev)} for _ in range(16)]
for mode in os.environ.get("MSM_MODES", "f16,f32").split(","):
    rgbd.set_precision(mode)
    pipe = ts.BatchedTwoStage(rgbd, 16, (H, W), confident_score=0.0, topk=False)
    for _ in range(2):
        out = pipe(samples)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pipe(samples)
    torch.cuda.synchronize()
    t_one = (time.perf_counter() - t0) / 5
    sink = lambda i, lab, ref, rows: None
    pipe.run([samples] * 4, consume=sink)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run([samples] * 12, consume=sink)
    torch.cuda.synchronize()
    t_two = (time.perf_counter() - t0) / 12
    print(f"{mode}: {len(out[2])} crops; one batch in flight {1e3 * t_one:.2f} ms, two in flight {1e3 * t_two:.2f} ms per 16 frames ({16 / t_two:.0f} frames/s)")
