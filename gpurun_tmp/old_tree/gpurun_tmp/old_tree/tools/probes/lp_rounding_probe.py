"""Which 16-bit storage rounding of the encoder's inter-kernel tensors costs how many mask bits (tuning aid, bf16 plan):
the round-3 kernels (fp32 tensors between the kernels) with value / attn / proj rounded to bf16 or fp16 in between, batch 8 at
640x480 against the fp32 reference golden.  python tools/probes/lp_rounding_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from unseenobjectswithmeanshift_amd import ops, synthetic as syn  # noqa: E402
import test_gpu_configs as tc  # noqa: E402

g = np.load(os.path.join(os.path.dirname(tc.__file__), "golden", "head_480x640_b8.npz"))
head = tc.make_head()
head.set_precision("bf16")
feats = {k: v.to("cuda") for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}


def score(tag):
    out, _ = head(feats)
    rates, ious = [], []
    for b in range(8):
        ref = tc.unpack(g[f"b{b}_sign_bits"], (100, 120, 160))
        gb = out["pred_masks"][b].cpu() > 0
        rates.append(float((gb != ref).float().mean()))
        inter, union = tc.iou_rows(gb, ref)
        ious.append((inter / union.clamp_min(1))[union >= 16])
    ious = torch.cat(ious)
    print(f"{tag:34s} mismatch mean {np.mean(rates):.4f} max {np.max(rates):.4f} | IoU mean {float(ious.mean()):.4f} min {float(ious.min()):.3f} "
          f"p01 {float(ious.quantile(0.01)):.3f} share>=0.9 {float((ious >= 0.9).float().mean()):.3f}", flush=True)


score("hm kernels (bf16 storage)")
head.pixel_decoder.hm_activations = False
score("round-3 kernels (fp32 storage)")
orig_gather, orig_block = ops.ms_deform_attn_encoder, ops.encoder_block_lp
rnd = {"bf16": lambda t: t.to(torch.bfloat16).float(), "fp16": lambda t: t.to(torch.float16).float(), None: lambda t: t}
for what in ("value", "attn", "proj", "all"):
    for dt in ("bf16", "fp16"):
        rv = rnd[dt if what in ("value", "all") else None]
        ra = rnd[dt if what in ("attn", "all") else None]
        rp = rnd[dt if what in ("proj", "all") else None]
        ops.ms_deform_attn_encoder = lambda value, ss, st, proj, h, p, rv=rv, ra=ra, rp=rp: ra(orig_gather(rv(value), ss, st, rp(proj), h, p))
        score(f"{what} stored as {dt}")
ops.ms_deform_attn_encoder = orig_gather
