"""3x3 FPN convolution (64 -> 64, B = 8, 120 x 160): one against two 16-pixel blocks per wave, fp32 and bf16 forms (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402
from unseenobjectswithmeanshift_amd._lib import set_option  # noqa: E402
from microbench import timeit_graph  # noqa: E402

DEV = "cuda:0"
B, H, W = 8, 120, 160
x = torch.randn(B, H * W, 64, device=DEV)
w3 = (torch.randn(64, 64, 3, 3, device=DEV) * 0.06).permute(0, 2, 3, 1).reshape(64, 576).contiguous()
res = {}
for wide in (0, 1):
    set_option("CONV3_WIDE", wide)
    for bf in (False, True):
        t = timeit_graph(lambda: ops.conv3x3_c64(x, w3, H, W, bf16=bf))
        res[(wide, bf)] = ops.conv3x3_c64(x, w3, H, W, bf16=bf)
        print(f"conv3x3_c64 {'bf16' if bf else 'fp32'}, {'two blocks' if wide else 'one block'} per wave: {t:.1f} us")
for bf in (False, True):
    print("equal outputs / moments:", torch.equal(res[(0, bf)][0], res[(1, bf)][0]), float((res[(0, bf)][1] - res[(1, bf)][1]).abs().max()))
