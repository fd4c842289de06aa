"""The decoder's batched K/V projection (nine jobs, B = 8, 640x480 levels) under HIP-graph timing (tuning aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import ops
from microbench import timeit_graph
DEV = "cuda:0"
B = 8
lv = [(15, 20), (30, 40), (60, 80)]
xs = [torch.randn(B, 64, h, w, device=DEV) for h, w in lv] * 3
ws = [torch.randn(512, 64, device=DEV) * 0.1 for _ in range(9)]
cs = [torch.randn(h * w, 512, device=DEV) for h, w in lv] * 3
t = timeit_graph(lambda: ops.kv_project_multi(xs, ws, cs))
mb = sum(B * h * w * 512 * 4 for h, w in lv) * 3 / 1e6
print(f"kv_project_multi: {t:.1f} us ({mb:.0f} MB written, {mb / t:.2f} TB/s; {2.0 * mb / 4 * 64 / t / 1e6 * 1e6 / 1e6:.1f} TFLOP/s)")
# separable constants (row + column tables instead of the per-position matrix)
cs2 = [torch.randn(h + w, 512, device=DEV) for h, w in lv] * 3
cw = [w for _, w in lv] * 3
for kw, name in ((dict(), "f32"), (dict(split=True), "f32_split"), (dict(out_dtype=torch.bfloat16), "bf16")):
    t1 = timeit_graph(lambda: ops.kv_project_multi(xs, ws, cs, **kw))
    t2 = timeit_graph(lambda: ops.kv_project_multi(xs, ws, cs2, cmat_widths=cw, **kw))
    print(f"kv_project_multi {name}: dense constants {t1:.1f} us, separable {t2:.1f} us")
