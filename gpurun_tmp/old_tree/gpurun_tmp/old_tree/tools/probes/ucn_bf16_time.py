"""UCN path (batch 2, 307 200 keys) in the three precision modes: time and agreement of the final masks with fp32 (tuning aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import _lib, synthetic as syn  # noqa: E402
from unseenobjectswithmeanshift_amd.meta_arch import PretrainedMeanShiftMaskFormer, build_ucn_head  # noqa: E402

dev = torch.device("cuda")
H, W, Q, UB = 480, 640, 100, 2
uh = build_ucn_head()
uh.pixel_decoder.load_state_dict(syn.synth_state_dict({"mask_features.weight": (256, 64, 3, 3), "mask_features.bias": (256,)}, salt=3))
uh.predictor.load_state_dict(syn.synth_state_dict(syn.decoder_param_shapes(dec_layers=6, num_feature_levels=1), salt=4))
ucn = PretrainedMeanShiftMaskFormer(backbone=None, sem_seg_head=uh.to(dev).eval(), num_queries=Q)
X, _ = syn.synth_unit_embeddings(H * W, 64, clusters=12, sigma=0.3, seed=5)
emb = X.view(1, H * W, 64).transpose(1, 2).reshape(1, 64, H, W).repeat(UB, 1, 1, 1).contiguous().to(dev)
ufe = {"res5": emb}
ref = None
for mode, fused in (("f32", True), ("bf16", False), ("bf16", True), ("f16", False), ("f16", True)):
    try:
        ucn.set_precision(mode)
        uh.predictor.fused_kv_attention = fused
        mode = f"{mode} fused_kv={fused}"
        for _ in range(2):
            out = ucn.inference(ufe, (H, W))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = ucn.inference(ufe, (H, W))
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 5
        with _lib.CallTimer() as ct:
            ucn.inference(ufe, (H, W))
            torch.cuda.synchronize()
        ud = ct.durations()
        top = {k: round(sum(v), 3) for k, v in sorted(ud.items(), key=lambda kv: -sum(kv[1]))[:5]}
        masks = out[2] if isinstance(out, (tuple, list)) else None
        msg = ""
        if ref is None:
            ref = [o.clone() if torch.is_tensor(o) else o for o in out]
        else:
            for i, (a, b) in enumerate(zip(out, ref)):
                if torch.is_tensor(a) and a.shape == b.shape and a.dtype.is_floating_point and a.numel() > 1000:
                    msg += f" out[{i}] mismatch of (x > 0.5) bits {float(((a > 0.5) != (b > 0.5)).float().mean()):.4%}, max |d| {float((a - b).abs().max()):.3g};"
        print(f"{mode}: {1e3 * t:.2f} ms per batch of {UB} = {UB / t:.1f} images/s; {top};{msg}")
    except Exception as e:  # noqa: BLE001
        print(f"{mode}: failed: {type(e).__name__}: {e}")
