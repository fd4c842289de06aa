"""Mean-shift unit timing (tuning aid): python tools/probes/meanshift_time.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import mean_shift as ms, ops, synthetic as syn  # noqa: E402
from microbench import timeit  # noqa: E402

H, W, S = 480, 640, 100
X, _ = syn.synth_unit_embeddings(H * W, 64, clusters=12, sigma=0.15, seed=3)
Xd = X.cuda()
feats = X.t().reshape(1, 64, H, W).contiguous().cuda()
print(f"seeding       {timeit(lambda: ops.ms_select_seeds(Xd, S, 7), iters=20):8.1f} us")
seeds, _ = ops.ms_select_seeds(Xd, S, 7)
print(f"hill climb    {timeit(lambda: ops.ms_hill_climb(Xd, seeds, 20.0, 10), iters=20):8.1f} us")
print(f"hill climb (f32_split) {timeit(lambda: ops.ms_hill_climb(Xd, seeds, 20.0, 10, precision='f32_split'), iters=20):8.1f} us")
Z = ops.ms_hill_climb(Xd, seeds, 20.0, 10)
print(f"components    {timeit(lambda: ops.ms_connected_components(Z, 0.04), iters=20):8.1f} us")
lab, num = ops.ms_connected_components(Z, 0.04)
print(f"assign        {timeit(lambda: ops.ms_assign(Xd, Z, lab, S), iters=20):8.1f} us")
np.random.seed(3)
for _ in range(3):
    ms.clustering_features(feats, num_seeds=S)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    ms.clustering_features(feats, num_seeds=S)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 50
print(f"clustering_features: {1e3 * t:.3f} ms = {1 / t:.1f} images/s")
for _ in range(3):
    ms.clustering_features(feats, num_seeds=S, precision="f32_split")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    ms.clustering_features(feats, num_seeds=S, precision="f32_split")
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 50
print(f"clustering_features (f32_split): {1e3 * t:.3f} ms = {1 / t:.1f} images/s")
