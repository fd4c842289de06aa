"""FPN lateral convolution (res2: 256 -> 64 at 120x160) + GroupNorm moments: tiled GEMM + moments pass against the input-projection kernel (tuning aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops, _lib
from microbench import timeit_graph
DEV = "cuda:0"
B, H, W = 8, 120, 160
x = torch.randn(B, 256, H, W, device=DEV)
w = torch.randn(64, 256, device=DEV) * 0.05
gw, gb = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
up = torch.randn(B, 60 * 80, 64, device=DEV)
def gemm_path():
    lat = ops.conv1x1_nchw_to_tokens(x, w, None)
    return ops.groupnorm_tokens(lat, gw, gb, H, W, groups=32, up=up, up_hw=(60, 80), eps=1e-5)
wp = ops.pack_conv_in_weight(w)
def in_path():
    lat, st = ops.conv1x1_in(x, wp, None)
    return ops.groupnorm_tokens(lat, gw, gb, H, W, groups=32, up=up, up_hw=(60, 80), eps=1e-5, stats=st, stats_ready=True)
print(f"gemm only           {timeit_graph(lambda: ops.conv1x1_nchw_to_tokens(x, w, None)):.1f} us")
print(f"gemm + GN (stats pass + apply) {timeit_graph(gemm_path):.1f} us")
for nt in (0,):
    with _lib.option("CONVIN_NT", nt):
        print(f"CONVIN_NT={nt}: conv1x1_in only {timeit_graph(lambda: ops.conv1x1_in(x, wp, None)):.1f} us;  + GN apply {timeit_graph(in_path):.1f} us")
a, b = gemm_path(), in_path()
print("max diff", float((a - b).abs().max()))
