"""Throughput of the head at 640x480, batch 8, against the number of batches in flight (graphs.PipelinedInference depth), per plan."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("MSM_QUEUES", "8"))
import torch  # noqa: E402

import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402
from unseenobjectswithmeanshift_amd.graphs import PipelinedInference  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
for mode in ("f16", "f32"):
    model.set_precision(mode)
    res = []
    for depth in (2, 3, 4, 5, 6, 8):
        pipe = PipelinedInference(model, depth=depth)
        for _ in range(depth):
            pipe.submit(feats, (480, 640))
        pipe.drain()
        run = lambda: pipe.submit(None, (480, 640), slot_inputs=True)
        for _ in range(4 * depth):
            run()
        pipe.drain()
        tp = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(400):
                run()
            torch.cuda.synchronize()
            tp.append((time.perf_counter() - t0) / 400)
        del pipe
        res.append(f"{depth}: {8 / min(tp):.0f}")
    print(f"{mode} (GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}) images/s by batches in flight: " + "  ".join(res))
