"""configs[4] clustering (n = 1 228 800, 300 seeds, 20 iterations) per phase and precision (tuning aid)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import ops, synthetic as syn, mean_shift as ms  # noqa: E402

n, S, iters = 960 * 1280, 300, 20
X, ids = syn.synth_unit_embeddings(n, 64, clusters=24, sigma=0.15, seed=3)
X = X.to("cuda")


def t(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(f"pack bf16: {t(lambda: ops.ms_pack_bf16(X)):.3f} ms")
xb = ops.ms_pack_bf16(X)
print(f"seeding fp32 stepwise: {t(lambda: ops.ms_select_seeds(X, S, 11)):.2f} ms")
print(f"seeding bf16 stepwise: {t(lambda: ops.ms_select_seeds(X, S, 11, xb=xb)):.2f} ms")
seeds, _ = ops.ms_select_seeds(X, S, 11)
for p in ("f32", "f32_split", "bf16"):
    print(f"hill climb {p}: {t(lambda: ops.ms_hill_climb(X, seeds, 20.0, iters, precision=p, xb=xb if p == 'bf16' else None)):.2f} ms")
for p in ("f32", "f32_split", "bf16"):
    print(f"mean_shift_smart_init {p}: {t(lambda: ms.mean_shift_smart_init(X, 20, S, iters, first_index=11, precision=p)):.2f} ms")
