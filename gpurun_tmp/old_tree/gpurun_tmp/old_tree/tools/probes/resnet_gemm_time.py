"""ResNet-50 backbone, batch 8 at 480x640: MIOpen convolutions everywhere against the 1x1 convolutions as plain GEMMs on the NHWC view
(hipBLASLt through torch.addmm / _addmm_activation), fp32 and bf16 (tuning aid)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd.resnet_backbone import ResNet50Backbone  # noqa: E402

dev = "cuda"
bb = ResNet50Backbone().to(dev).eval()
images = torch.randn(8, 3, 480, 640, device=dev)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def conv1x1(x, w2d, b, relu, stride=1):
    """x (B, C, H, W) channels_last; w2d (Cout, Cin); -> (B, Cout, H', W') channels_last, through one GEMM on the NHWC view"""
    if stride != 1:
        x = x[:, :, ::stride, ::stride].contiguous(memory_format=torch.channels_last)
    B, C, H, W = x.shape
    a = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
    y = torch._addmm_activation(b, a, w2d.t(), use_gelu=False) if relu else torch.addmm(b, a, w2d.t())
    return y.view(B, H, W, -1).permute(0, 3, 1, 2)


def run_gemm(plan, x):
    (ws, bs), stages = plan
    x = x.to(ws.dtype).contiguous(memory_format=torch.channels_last)
    x = F.max_pool2d(F.relu(F.conv2d(x, ws, bs, stride=2, padding=3)), 3, stride=2, padding=1)
    out = []
    for blocks in stages:
        for (w1, b1), (w2, b2), (w3, b3), sc, stride in blocks:
            y = conv1x1(x, w1.flatten(1), b1, True)
            y = F.relu(F.conv2d(y, w2, b2, stride=stride, padding=1))
            y = conv1x1(y, w3.flatten(1), b3, False)
            x = F.relu(y + (x if sc is None else conv1x1(x, sc[0].flatten(1), sc[1], False, stride[0] if isinstance(stride, tuple) else stride)))
        out.append(x)
    return out


for mode in ("f32", "bf16"):
    bb.backbone_dtype = mode
    plan = bb._plan()
    ref = bb(images)
    got = run_gemm(plan, images)
    err = max(float((a.float() - b).abs().max() / b.abs().max()) for a, b in zip(got, ref.values()))
    print(f"{mode}: MIOpen everywhere {timed(lambda: bb(images)):.2f} ms; 1x1 as GEMMs {timed(lambda: run_gemm(plan, images)):.2f} ms (max rel diff {err:.2e})", flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run_gemm(plan, images)
        with torch.cuda.graph(g, stream=s):
            run_gemm(plan, images)
    torch.cuda.synchronize()
    print(f"   1x1 as GEMMs, HIP graph: {timed(g.replay):.2f} ms")
