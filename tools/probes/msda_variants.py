"""MSDeformAttn gather variants on one GPU: bitwise comparison and HIP-graph timing (tuning aid).
   python tools/probes/msda_variants.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import _lib, ops  # noqa: E402
from microbench import timeit_graph  # noqa: E402

DEV = "cuda"
B, S = 8, 6300
torch.manual_seed(0)
value, proj = torch.randn(B, S, 64, device=DEV), torch.randn(B, S, 288, device=DEV) * 0.3
ss = torch.tensor([(15, 20), (30, 40), (60, 80)], dtype=torch.int64, device=DEV)
st = torch.tensor([0, 300, 1500], dtype=torch.int64, device=DEV)
vhm = ops.value_to_head_major(value, 8)
outs = {}
for name, opt in (("owner records (default)", _lib.OPT_AUTO), ("round-2 kernel", 2), ("8x8 workgroups", 3), ("generic", 1)):
    with _lib.option("MSDA_GENERIC", opt):
        outs[name] = ops.ms_deform_attn_encoder(vhm, ss, st, proj, 8, 4).clone()
        t = timeit_graph(lambda: ops.ms_deform_attn_encoder(vhm, ss, st, proj, 8, 4))
    print(f"{name:28s} {t:7.1f} us", flush=True)
ref = outs["round-2 kernel"]
for name, o in outs.items():
    print(f"{name:28s} equal to round-2 kernel: {torch.equal(o, ref)}  max|d| {float((o - ref).abs().max()):.2e}")
# large offsets (many taps out of range) and a short map
for scale, Bx in ((3.0, 2), (0.05, 1)):
    p2 = torch.randn(Bx, S, 288, device=DEV) * scale
    v2 = ops.value_to_head_major(torch.randn(Bx, S, 64, device=DEV), 8)
    a = ops.ms_deform_attn_encoder(v2, ss, st, p2, 8, 4)
    with _lib.option("MSDA_GENERIC", 2):
        b = ops.ms_deform_attn_encoder(v2, ss, st, p2, 8, 4)
    print(f"offset scale {scale}: equal {torch.equal(a, b)}")

# ---- fused sampling projection: gather + the token kernel with and without the proj tail ----
from microbench import timeit  # noqa: E402
src, pos = torch.randn(B, S, 64, device=DEV), torch.randn(S, 64, device=DEV)
wp, bp = torch.randn(288, 64, device=DEV) * 0.04, torch.randn(288, device=DEV) * 0.3
wpack, bpack = ops.pack_msda_proj(wp, bp, 8, 3, 4)
t = timeit_graph(lambda: ops.ms_deform_attn_encoder_fused(vhm, ss, st, src, pos, wpack, bpack, 4))
print(f"fused projection + gather      {t:7.1f} us", flush=True)
attn = torch.randn(B, S, 64, device=DEV)
wo, w1, w2 = torch.randn(64, 64, device=DEV) * .1, torch.randn(1024, 64, device=DEV) * .1, torch.randn(64, 1024, device=DEV) * .03
wv = torch.randn(64, 64, device=DEV) * .1
stream = ops.pack_encoder_block(wo, w1, w2, wv, wp)
small = torch.randn(64 * 7 + 1024 + 288, device=DEV) * .1
for pw in (288, 0):
    t = timeit_graph(lambda: ops.encoder_block(attn, src, stream, small, 1024, pw, pos=pos, tokens_per_image=S, value_heads=8), reps=10)
    print(f"enc_block proj_width={pw:3d}       {t:7.1f} us", flush=True)
