#!/usr/bin/env python
"""Kernel micro-benchmarks on one GPU (tuning aid, not part of the product or the tests).
   python tools/microbench.py gemm | enc | attn | mask"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3   # us


def gemm():
    shapes = [("ffn1", 50400, 1024, 64), ("ffn2", 50400, 64, 1024), ("val", 50400, 64, 64), ("proj", 50400, 288, 64),
              ("kv2", 38400, 256, 256), ("kv1", 9600, 256, 256), ("q", 800, 256, 256), ("dffn1", 800, 2048, 256)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(N, K, device=DEV) * K ** -0.5
        b = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        t = timeit(lambda: ops.gemm(a, w, b, out=out))
        print(f"{name:6s} M={M:6d} N={N:5d} K={K:5d}  {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s  "
              f"tile={os.environ.get('MSM_GEMM_TILE', 'auto')} nostore={'MSM_GEMM_NOSTORE' in os.environ}", flush=True)


if __name__ == "__main__":
    {"gemm": gemm}[sys.argv[1]]()
