"""One batch in flight (B = 8, 640x480) for the FFN tail's hidden-slice counts (tuning aid)."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
for prec in ("f32", "bf16"):
    model.set_precision(prec)
    for parts in (None, 8, 4, 2):
        model.sem_seg_head.predictor.ffn_parts = parts
        p = model.pipelined(depth=1)
        p.submit(feats, (480, 640)); p.drain()
        run = lambda: p.submit(None, (480, 640), slot_inputs=True)
        for _ in range(5):
            run()
        p.drain()
        t = bench.timed(run, 200)
        print(f"{prec} ffn_parts={parts}: one batch in flight {1e3 * t:.3f} ms", flush=True)
        del p
