"""3x3 FPN convolution at B=8, 120x160: fp32 MFMA / bf16 / f32_split forms and the GroupNorm that feeds them (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402
from microbench import timeit_graph  # noqa: E402

B, H, W = 8, 120, 160
x = torch.randn(B, H * W, 64, device="cuda")
up = torch.randn(B, (H // 2) * (W // 2), 64, device="cuda")
w3 = (torch.randn(64, 576, device="cuda") * 0.04).contiguous()
g, be = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
st = ops.groupnorm_stats(x)
kw = dict(up=up, up_hw=(H // 2, W // 2), stats=st, stats_ready=True)
y = ops.groupnorm_tokens(x, g, be, H, W, **kw)
pl = ops.groupnorm_tokens(x, g, be, H, W, split_planes=True, **kw)
s0 = torch.zeros(B, 64, 2, device="cuda", dtype=torch.float64)
print(f"groupnorm fp32 out   {timeit_graph(lambda: ops.groupnorm_tokens(x, g, be, H, W, **kw)):7.1f} us")
print(f"groupnorm 3 planes   {timeit_graph(lambda: ops.groupnorm_tokens(x, g, be, H, W, split_planes=True, **kw)):7.1f} us")
print(f"conv3x3 fp32 MFMA    {timeit_graph(lambda: ops.conv3x3_c64(y, w3, H, W, stats=s0, stats_cleared=True)):7.1f} us")
print(f"conv3x3 bf16         {timeit_graph(lambda: ops.conv3x3_c64(y, w3, H, W, stats=s0, stats_cleared=True, bf16=True)):7.1f} us")
print(f"conv3x3 f32_split    {timeit_graph(lambda: ops.conv3x3_c64(pl, w3, H, W, stats=s0, stats_cleared=True, split=True)):7.1f} us")
