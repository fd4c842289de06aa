"""The f32_split hill climb alone (for rocprofv3 counter passes): python tools/probes/hill_split_only.py [fallback]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import _lib, ops, synthetic as syn  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "fallback":
    _lib.set_option("MS_SPLIT_KERNEL", 1)
X, _ = syn.synth_unit_embeddings(480 * 640, 64, clusters=12, sigma=0.15, seed=3)
Xd = X.cuda()
seeds, _ = ops.ms_select_seeds(Xd, 100, 7)
for _ in range(3):
    ops.ms_hill_climb(Xd, seeds, 20.0, 10, precision="f32_split")
torch.cuda.synchronize()
