"""clustering_features on a batch of B = 8 maps of 640x480 (planted clusters), a few times -- under rocprofv3 --kernel-trace --stats
(tools/probes/stats_table.py prints the per-kernel table); prints the wall time per image without the profiler's help."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import mean_shift as ms, synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
H, W, B = 480, 640, int(os.environ.get("MS_B", "8"))
maps = []
for j in range(B):
    X, _ = syn.synth_unit_embeddings(H * W, 64, clusters=12, sigma=0.15, seed=3 + j, background_frac=0.02 if os.environ.get("MS_NOISY") else 0.0)
    maps.append(X.t().reshape(64, H, W))
feats = torch.stack(maps).contiguous().to(dev)
np.random.seed(3)
fn = getattr(ms, os.environ.get("MS_FN", "clustering_features"))
for _ in range(3):
    fn(feats, num_seeds=100)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = int(os.environ.get("MS_REPS", "5"))
for _ in range(reps):
    fn(feats, num_seeds=100)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / reps
print(f"{fn.__name__}: B = {B}: {1e3 * t:.3f} ms per batch, {1e3 * t / B:.3f} ms per image, {B / t:.1f} images/s")
