"""Per-entry-point kernel time of one pass of the head (batch 8, 640x480) in the given precisions, eager with HIP events around every
library call (_lib.CallTimer), plus the HIP-graph time of the pass:  python tools/probes/plan_breakdown.py [bf16 f16 ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from unseenobjectswithmeanshift_amd import _lib, synthetic as syn  # noqa: E402
import test_gpu_configs as tc  # noqa: E402
from microbench import timeit_graph  # noqa: E402

head = tc.make_head()
feats = {k: v.cuda() for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
for mode in (sys.argv[1:] or ["bf16", "f16"]):
    if ":" in mode:                                   # "f16:4096" = fused K/V attention from 4096 keys on
        mode, mk = mode.split(":")
        head.predictor.fused_kv_min_keys = int(mk)
    head.set_precision(mode)
    for _ in range(3):
        head(feats)
    with _lib.CallTimer() as ct:
        for _ in range(5):
            head(feats)
        torch.cuda.synchronize()
    dur = ct.durations()
    tot = sum(sum(v) for v in dur.values()) / 5
    print(f"== {mode}: graph {timeit_graph(lambda: head(feats)):.1f} us per pass; sum of kernels {1e3 * tot:.1f} us")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"   {k:44s} {len(v) // 5:3d} x {1e3 * sum(v) / len(v):7.1f} us = {1e3 * sum(v) / 5:7.1f} us")
