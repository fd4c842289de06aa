"""bf16-plan encoder block timing at configs[1] size (tuning aid): python tools/probes/enc_lp_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402
from microbench import timeit_graph  # noqa: E402

DEV = "cuda"
B, S = 8, 6300
torch.manual_seed(0)
attn, src, pos = torch.randn(B, S, 64, device=DEV), torch.randn(B, S, 64, device=DEV), torch.randn(S, 64, device=DEV)
wo, w1, w2 = torch.randn(64, 64, device=DEV) * .1, torch.randn(1024, 64, device=DEV) * .1, torch.randn(64, 1024, device=DEV) * .03
wv, wp = torch.randn(64, 64, device=DEV) * .1, torch.randn(288, 64, device=DEV) * .1
small = torch.randn(64 * 7 + 1024 + 288, device=DEV) * .1
for name, pack, fn in (("lp", ops.pack_encoder_block_lp, ops.encoder_block_lp), ("split", ops.pack_encoder_block_split, ops.encoder_block_split)):
    stream = pack(wo, w1, w2, wv, wp)
    t = timeit_graph(lambda: fn(attn, src, stream, small, 1024, 288, pos=pos, tokens_per_image=S, value_heads=8))
    print(f"enc_block {name}: {t:7.1f} us", flush=True)

# round 4: head-major bf16 kernels (csrc/enc_lp.hip)
bp = torch.randn(288, device=DEV) * 0.5
wp2 = wp * 0.3
stream = ops.pack_encoder_block_hm(wo, w1, w2, wv, wp2)
stream_last = ops.pack_encoder_block_hm(wo, w1, w2)
sl = lambda i: small[i * 64:(i + 1) * 64]
sm = ops.pack_encoder_block_hm_small(sl(0), sl(1), sl(2), small[192:192 + 1024], small[1216:1280], small[1280:1344], small[1344:1408], small[1408:1472], bp)
attn_hm = torch.randn(B, 8, S, 8, device=DEV).to(torch.float16)
t = timeit_graph(lambda: ops.encoder_block_hm(attn_hm, src, stream, sm, 1024, pos=pos))
print(f"enc_block hm: {t:7.1f} us", flush=True)
t = timeit_graph(lambda: ops.encoder_block_hm(attn_hm, src, stream_last, sm, 1024, want_next=False))
print(f"enc_block hm (last layer): {t:7.1f} us", flush=True)
ss = torch.tensor([(15, 20), (30, 40), (60, 80)], dtype=torch.int64, device=DEV)
st = torch.tensor([0, 300, 1500], dtype=torch.int64, device=DEV)
wpack, bpack = ops.pack_msda_proj_lp(wp2, bp)
value_hm = torch.randn(B, 8, S, 8, device=DEV).to(torch.float16)
proj = torch.nn.functional.linear(src + pos, wp2, bp).contiguous()
proj_hm = ops.proj_to_head_major_records(proj)
t = timeit_graph(lambda: ops.ms_deform_attn_encoder_lp(value_hm, ss, st, proj_hm, 4))
print(f"msda enc lp (stored bf16 projection, bf16 taps): {t:7.1f} us", flush=True)
t = timeit_graph(lambda: ops.ms_deform_attn_encoder_lp_fused(value_hm, ss, st, src, pos, wpack, bpack, 4))
print(f"msda enc lp (fused projection, bf16 taps): {t:7.1f} us", flush=True)
vf = value_hm.float()
t = timeit_graph(lambda: ops.ms_deform_attn_encoder(vf, ss, st, proj, 8, 4))
print(f"msda enc fp32 (rec kernel): {t:7.1f} us", flush=True)
t = timeit_graph(lambda: ops.to_f16(vf))
print(f"to_f16 (B,8,S,8): {t:7.1f} us", flush=True)
