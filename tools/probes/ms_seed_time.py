"""Seeding over the bf16 copy: persistent (on-chip + streamed tail) against one launch per step, by map size (tuning aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import ops, synthetic as syn  # noqa: E402

S = 300
for n in (655360, 917504, 1228800, 1600000):
    X, _ = syn.synth_unit_embeddings(n, 64, clusters=24, sigma=0.15, seed=3)
    X = X.to("cuda")
    xb = ops.ms_pack_bf16(X)

    def t(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    tp = t(lambda: ops.ms_select_seeds(X, S, 11, xb=xb))
    ts = t(lambda: ops.ms_select_seeds(X, S, 11, xb=xb, stepwise=True))
    print(f"n={n}: persistent {tp:.2f} ms ({tp / (S - 1) * 1e3:.1f} us/step), stepwise {ts:.2f} ms")
