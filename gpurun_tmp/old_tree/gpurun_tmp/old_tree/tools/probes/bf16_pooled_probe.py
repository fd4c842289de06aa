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


def v_f16():
    """The 16-bit plan with IEEE-half operands where the range is bounded (head.set_precision("f16")), whole and part by part."""
    head.set_precision("f16")
    score('f16 plan (set_precision("f16"))')
    pred, pd = head.predictor, head.pixel_decoder
    head.set_precision("f32")
    pd.precision, pd.lp_operands = "bf16", "f16"
    score("only encoder 16-bit, FFN on fp16 operands")
    head.set_precision("f32")
    pred.tails_dtype = "f16"
    score("only tails 16-bit, fp16 operands")
    head.set_precision("f32")
    pred.attention_dtype = "bf16"
    if hasattr(pred, "attention_keys"):
        pred.attention_keys = "f16"
    score("only attention 16-bit, fp16 keys / scores")
    head.set_precision("f32")


def v_encoder_parts():
    """Only the pixel decoder in the 16-bit plan (fp16 FFN operands), with one of its low-precision pieces switched back to fp32."""
    pd = head.pixel_decoder
    head.set_precision("f32")
    pd.precision, pd.lp_operands = "bf16", "f16"
    score("encoder 16-bit (fp16 FFN): all pieces")
    for attr in ("lp_conv3x3", "lp_input_proj", "lp_prologue", "hm_activations"):
        setattr(pd, attr, False)
        score(f"  ... with {attr} = False")
        setattr(pd, attr, True)
    pd.lp_conv3x3 = pd.lp_input_proj = pd.lp_prologue = False
    score("  ... encoder layers only (no lp conv3x3 / lateral / prologue)")
    pd.lp_conv3x3 = pd.lp_input_proj = pd.lp_prologue = True
    head.set_precision("f32")


def v_sensitivity():
    """(i) How far the pixel decoder's outputs are from their fp32 values in each 16-bit form (relative L2 error of the encoder tokens
    and of the 64-channel mask activation); (ii) what a synthetic relative perturbation eps of those outputs costs in mask bits with
    EVERYTHING else fp32 -- the response curve of the head to encoder error."""
    pd = head.pixel_decoder
    head.set_precision("f32")
    feats = FEATS[10]
    ref = pd.forward_features(feats, folded=True)
    r_src = torch.cat([t.flatten(1) for t in ref[2]], 1).clone()
    r_act = ref[0].act.clone()

    def err(tag):
        got = pd.forward_features(feats, folded=True)
        g_src = torch.cat([t.flatten(1) for t in got[2]], 1)
        e1 = float((g_src - r_src).norm() / r_src.norm())
        e2 = float((got[0].act - r_act).norm() / r_act.norm())
        print(f"{tag:46s} rel L2 error: encoder tokens {e1:.2e}, mask activation {e2:.2e}", flush=True)

    pd.precision, pd.lp_operands = "bf16", "bf16"
    err("encoder bf16 plan (bf16 FFN)")
    pd.lp_operands = "f16"
    err("encoder 16-bit plan (fp16 FFN)")
    pd.hm_activations = False
    err("round-3 kernels (fp32 storage, bf16 FFN)")
    pd.hm_activations = True
    pd.precision = "f32_split"
    err("f32_split")
    head.set_precision("f32")
    # storage roundings alone: the fp32 encoder with value / attn / proj rounded to 16 bits between its kernels
    og = ops.ms_deform_attn_encoder
    rnd = {"bf16": lambda t: t.to(torch.bfloat16).float(), "fp16": lambda t: t.to(torch.float16).float(), None: lambda t: t}
    pd.fused_msda = False
    for what in ("value", "attn", "proj", "all"):
        for dt in ("fp16", "bf16"):
            rv, ra, rp = (rnd[dt if what in (k, "all") else None] for k in ("value", "attn", "proj"))
            ops.ms_deform_attn_encoder = lambda value, ss, st, proj, h, p, rv=rv, ra=ra, rp=rp: ra(og(rv(value), ss, st, rp(proj), h, p))
            err(f"fp32 encoder, {what} rounded to {dt} between kernels")
    def part(lo, hi):
        def f(t):
            t = t.clone()
            t[..., lo:hi] = t[..., lo:hi].to(torch.float16).float()
            return t
        return f
    seen = {}
    def stats(value, ss, st, proj, h, p):
        if not seen:
            off = proj[..., :192].abs()
            lg = proj[..., 192:]
            seen["done"] = 1
            print(f"  sampling offsets |o|: mean {float(off.mean()):.2f} px, 90 % {float(off.flatten()[::97].quantile(0.9)):.2f}, max {float(off.max()):.2f}; "
                  f"attention logits: std {float(lg.std()):.2f}, max |l| {float(lg.abs().max()):.2f}", flush=True)
        return og(value, ss, st, proj, h, p)
    ops.ms_deform_attn_encoder = stats
    err("fp32 encoder (statistics of the projection)")
    for tag, f in (("offsets only -> fp16", part(0, 192)), ("logits only -> fp16", part(192, 288))):
        ops.ms_deform_attn_encoder = lambda value, ss, st, proj, h, p, f=f: og(value, ss, st, f(proj), h, p)
        err(f"fp32 encoder, proj {tag}")
    ops.ms_deform_attn_encoder = og
    if "quick" in sys.argv:
        return
    orig = pd.forward_features
    for eps in (1e-5, 1e-4, 3e-4, 1e-3, 3e-3):
        def noisy(features, folded=False, eps=eps):
            mf, o0, ms = orig(features, folded=folded)
            g = torch.Generator(device="cuda").manual_seed(1)
            for t in ms:                                   # views of ONE token buffer: perturb in place
                t.mul_(1.0 + eps * torch.randn(t.shape, device=t.device, generator=g))
            mf.act.mul_(1.0 + eps * torch.randn(mf.act.shape, device=mf.act.device, generator=g))
            return mf, o0, ms
        pd.forward_features = noisy
        score(f"fp32 plan, encoder outputs x (1 + {eps:g} N(0,1))")
    pd.forward_features = orig


VARIANTS = {"default": v_default, "round3": v_round3, "f32": v_f32, "parts": v_parts, "storage": v_storage, "f16": v_f16, "encoder_parts": v_encoder_parts, "sensitivity": v_sensitivity}

if __name__ == "__main__":
    for name in ([a for a in sys.argv[1:] if a != "quick"] or list(VARIANTS)):
        VARIANTS[name]()
