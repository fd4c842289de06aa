"""bf16 plan against the fp32 reference goldens of FOUR batches of 8 (input seeds 10..13; 3200 masks pooled): which part of the
low-precision plan costs how many mask bits.  Single chaotic events average out over 32 images; arithmetic shows.

    python tools/probes/bf16_pooled_probe.py [variant ...]        (no argument: every variant)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from unseenobjectswithmeanshift_amd import ops, synthetic as syn  # noqa: E402
import test_gpu_configs as tc  # noqa: E402

SEEDS = (10, 11, 12, 13)
GOLD = {s: np.load(os.path.join(ROOT, "tests", "golden", "head_480x640_b8.npz" if s == 10 else f"head_480x640_b8_s{s}.npz")) for s in SEEDS}
FEATS = {s: {k: v.to("cuda") for k, v in syn.synth_backbone_features(8, 480, 640, seed=s).items()} for s in SEEDS}
head = tc.make_head()


def score(tag):
    rates, ious, dmax = [], [], []
    for s in SEEDS:
        g = GOLD[s]
        out, _ = head(FEATS[s])
        idx = torch.from_numpy(g["mask_sample_idx"])
        for b in range(8):
            ref = tc.unpack(g[f"b{b}_sign_bits"], (100, 120, 160))
            pm = out["pred_masks"][b].cpu()
            gb = pm > 0
            rates.append(float((gb != ref).float().mean()))
            inter, union = tc.iou_rows(gb, ref)
            ious.append((inter / union.clamp_min(1))[union >= 16])
            dm = (pm.flatten()[idx] - torch.from_numpy(g[f"b{b}_sample_val"])).abs()
            dmax.append(float(dm.max()) / float(g["mask_absmax"][b]))
    ious = torch.cat(ious)
    print(f"{tag:46s} mismatch mean {100 * np.mean(rates):.3f} % max {100 * np.max(rates):.2f} % | IoU mean {float(ious.mean()):.4f} min {float(ious.min()):.3f} "
          f"p01 {float(ious.quantile(0.01)):.3f} >=0.9 {float((ious >= 0.9).float().mean()):.3f} | max|dmask|/range worst image {max(dmax):.3f} "
          f"median {np.median(dmax):.3f}", flush=True)


def v_default():
    head.set_precision("bf16")
    score("bf16 plan (default: hm kernels, fp16 storage)")


def v_round3():
    head.set_precision("bf16")
    head.pixel_decoder.hm_activations = False
    score("bf16 plan, round-3 encoder kernels (fp32 storage)")
    head.pixel_decoder.hm_activations = True


def v_parts():
    """One part of the plan in bf16 at a time (the rest fp32), then all but one."""
    pred, pd = head.predictor, head.pixel_decoder
    names = ("encoder", "tails", "attention", "mask_step")

    def apply(on):
        head.set_precision("f32")
        pd.precision = "bf16" if "encoder" in on else "f32"
        pred.tails_dtype = "bf16" if "tails" in on else "f32"
        pred.attention_dtype = "bf16" if "attention" in on else "f32"
        pred.mask_step_dtype = "bf16" if "mask_step" in on else "f32"

    for n in names:
        apply({n})
        score(f"only {n} in bf16")
    for n in names:
        apply(set(names) - {n})
        score(f"all but {n} in bf16")
    head.set_precision("f32")


def v_storage():
    """Round-3 kernels (fp32 tensors between the encoder kernels) with ONE of the three rounded to fp16 / bf16 in between."""
    head.set_precision("bf16")
    head.pixel_decoder.hm_activations = False
    orig = ops.ms_deform_attn_encoder
    rnd = {"bf16": lambda t: t.to(torch.bfloat16).float(), "fp16": lambda t: t.to(torch.float16).float(), None: lambda t: t}
    try:
        for what in ("value", "attn", "proj", "all"):
            for dt in ("fp16", "bf16"):
                rv = rnd[dt if what in ("value", "all") else None]
                ra = rnd[dt if what in ("attn", "all") else None]
                rp = rnd[dt if what in ("proj", "all") else None]
                ops.ms_deform_attn_encoder = lambda value, ss, st, proj, h, p, rv=rv, ra=ra, rp=rp: ra(orig(rv(value), ss, st, rp(proj), h, p))
                score(f"round-3 kernels, {what} stored as {dt}")
    finally:
        ops.ms_deform_attn_encoder = orig
        head.pixel_decoder.hm_activations = True


def v_f32():
    head.set_precision("f32")
    score("fp32 plan (chaos floor of the instrument)")
    head.set_precision("f32_split")
    score("f32_split plan")
    head.set_precision("f32")


def v_tails_f16():
    """The decoder tails with fp16 instead of bf16 weights (MSM_OPT_LP_F16; same bytes, same MFMA rate): alone and inside the plan.
    A fresh head per setting: the packed weights are cached per parameter version, not per option."""
    global head
    from unseenobjectswithmeanshift_amd import _lib
    keep = head
    for opt, label in ((1, "fp16 weights, hi + lo fp16 activations"), (2, "fp16 weights, one fp16 activation term")):
        with _lib.option("LP_F16", opt):
            head = tc.make_head()
            head.set_precision("f32")
            head.predictor.tails_dtype = "bf16"
            score(f"only tails 16-bit: {label}")
            head.set_precision("bf16")
            score(f"whole plan, tails: {label}")
    head = keep


VARIANTS = {"default": v_default, "round3": v_round3, "f32": v_f32, "parts": v_parts, "storage": v_storage, "tails_f16": v_tails_f16}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(VARIANTS)):
        VARIANTS[name]()
