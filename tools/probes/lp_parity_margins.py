"""Margins of test_config2_slice_bf16_vs_reference per image, for the hm kernels and the round-3 kernels (tuning aid)."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402
import test_gpu_configs as tc  # noqa: E402

g = np.load(os.path.join(R, "tests", "golden", "head_480x640_b8.npz"))
head = tc.make_head()
head.set_precision("bf16")
feats = {k: v.to("cuda") for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
for hm in (True, False):
    head.pixel_decoder.hm_activations = hm
    out, _ = head(feats)
    ious = []
    for b in range(8):
        ref = tc.unpack(g[f"b{b}_sign_bits"], (100, 120, 160))
        gb = out["pred_masks"][b].cpu() > 0
        rate = float((gb != ref).float().mean())
        inter, union = tc.iou_rows(gb, ref)
        iou = (inter / union.clamp_min(1))[union >= 16]
        ious.append(iou)
        scale = float(tc.T(g["mask_absmax"])[b])
        dm = (out["pred_masks"][b].cpu().flatten()[tc.T(g["mask_sample_idx"])] - tc.T(g[f"b{b}_sample_val"])).abs()
        print(f"hm={hm} image {b}: mismatch {rate:.4f}  dm mean/scale {float(dm.mean()) / scale:.4f}  dm max/scale {float(dm.max()) / scale:.4f}  "
              f"IoU mean {float(iou.mean()):.4f} share>=0.9 {float((iou >= 0.9).float().mean()):.3f}")
    ious = torch.cat(ious)
    print(f"hm={hm} all: min {float(ious.min()):.3f} p01 {float(ious.quantile(0.01)):.3f} mean {float(ious.mean()):.4f} share>=0.9 {float((ious >= 0.9).float().mean()):.3f}")
