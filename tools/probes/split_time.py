"""f32_split plan timing (tuning aid): python tools/probes/split_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from unseenobjectswithmeanshift_amd import ops, synthetic as syn  # noqa: E402
from microbench import timeit_graph  # noqa: E402

dev = torch.device("cuda", 0)
B, Q, H, W = 8, 100, 120, 160
wide = torch.randn(B, Q, 256, device=dev) * 0.3
f = torch.randn(B, 64, H, W, device=dev)
packed = ops.pack_mask_features_split(f)
print(f"pack split: {timeit_graph(lambda: ops.pack_mask_features_split(f)):.1f} us")
for tgt in ((15, 20), (30, 40), (60, 80), None):
    t0 = timeit_graph(lambda: ops.mask_logits(wide[..., :64], f, want_mask=tgt is None, target_size=tgt, qbias=wide[..., 64]))
    t1 = timeit_graph(lambda: ops.mask_logits(wide[..., :64], f, want_mask=tgt is None, target_size=tgt, qbias=wide[..., 64], packed_split=packed))
    print(f"mask step target {tgt}: fp32 MFMA {t0:.1f} us, split {t1:.1f} us")
model = bench.build_model(dev)
feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
for mode in ("f32", "f32_split"):
    model.set_precision(mode)
    p = model.pipelined(depth=1)
    p.submit(feats, (480, 640)); p.drain()
    run = lambda: p.submit(None, (480, 640), slot_inputs=True)
    for _ in range(5):
        run()
    p.drain()
    t = bench.timed(run, 200)
    print(f"{mode}: one batch in flight {1e3 * t:.3f} ms = {8 / t:.0f} images/s")
    del p
