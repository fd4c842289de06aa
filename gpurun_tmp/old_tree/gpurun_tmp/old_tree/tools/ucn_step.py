"""The UCN path (bench.py's `ucn_path` workload: batch 2 of 480x640 64-channel embeddings, 307 200 keys, 6 decoder layers, post-processing)
as a plain loop of eager passes -- the command the UCN per-kernel tables under profiles/ are collected with:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/<tag>_ucn -o <tag> -- python tools/ucn_step.py --precision bf16 --steps 10
    python tools/summarize_ucn_profile.py gpurun_out/<tag>_ucn <tag> bf16
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402
from unseenobjectswithmeanshift_amd.meta_arch import PretrainedMeanShiftMaskFormer, build_ucn_head  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16", choices=("f32", "bf16", "f16"))
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--unfused", action="store_true", help="K/V projection + attention as two launches (predictor.fused_kv_attention = False)")
a = ap.parse_args()
dev = torch.device("cuda")
H, W, Q, UB = 480, 640, 100, 2
uh = build_ucn_head()
uh.pixel_decoder.load_state_dict(syn.synth_state_dict({"mask_features.weight": (256, 64, 3, 3), "mask_features.bias": (256,)}, salt=3))
uh.predictor.load_state_dict(syn.synth_state_dict(syn.decoder_param_shapes(dec_layers=6, num_feature_levels=1), salt=4))
ucn = PretrainedMeanShiftMaskFormer(backbone=None, sem_seg_head=uh.to(dev).eval(), num_queries=Q)
X, _ = syn.synth_unit_embeddings(H * W, 64, clusters=12, sigma=0.3, seed=5)
emb = X.view(1, H * W, 64).transpose(1, 2).reshape(1, 64, H, W).repeat(UB, 1, 1, 1).contiguous().to(dev)
ucn.set_precision(a.precision)
uh.predictor.fused_kv_attention = not a.unfused
for _ in range(a.warmup):
    ucn.inference({"res5": emb}, (H, W))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    torch.cuda._sleep(1000)                  # marker launch (spin_kernel): tools/summarize_ucn_profile.py cuts the trace there
    ucn.inference({"res5": emb}, (H, W))
    torch.cuda.synchronize()                 # one pass at a time
t = (time.perf_counter() - t0) / a.steps
print(f"ucn_step precision={a.precision} fused_kv={not a.unfused}: {1e3 * t:.3f} ms per batch of {UB} (eager, synchronised per pass) = {UB / t:.1f} images/s")
