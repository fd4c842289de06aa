"""Batch size x batches in flight (tuning aid): how 8 images are best put through the chip.  python tools/probes/batch_split.py"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
for b, depth in ((8, 1), (4, 2), (2, 4), (8, 4), (4, 8), (16, 2), (16, 4), (32, 2)):
    feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(b, 480, 640, seed=10).items()}
    p = model.pipelined(depth=depth)
    for _ in range(depth):
        p.submit(feats, (480, 640))
    p.drain()
    run = lambda: p.submit(None, (480, 640), slot_inputs=True)
    for _ in range(3 * depth):
        run()
    p.drain()
    t = bench.timed(run, 60 * depth)
    print(f"batch {b:2d} x {depth} in flight: {1e3 * t:.3f} ms per batch, {b / t:.0f} images/s, {1e3 * t * depth:.3f} ms per round of {b * depth}", flush=True)
    del p
