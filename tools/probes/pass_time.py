"""Graph-replayed pass of the head at 640x480, batch 8, per precision plan -- from the package under $MSM_TREE when given (A/B of two
builds on one box): one batch in flight (GraphedInference, input copy included) and four in flight (PipelinedInference)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tree = os.environ.get("MSM_TREE")
if tree:
    sys.path.insert(0, os.path.abspath(tree))
sys.path.append(ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402
import unseenobjectswithmeanshift_amd  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402
from unseenobjectswithmeanshift_amd.graphs import PipelinedInference  # noqa: E402

dev = torch.device("cuda", 0)
if os.environ.get("MSM_OPTION"):                      # e.g. MSM_OPTION=ENC_NO_COOP=2
    from unseenobjectswithmeanshift_amd import _lib
    name, val = os.environ["MSM_OPTION"].split("=")
    _lib.set_option(name, int(val))
model = bench.build_model(dev)
feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
out = [os.path.dirname(unseenobjectswithmeanshift_amd.__file__).replace(ROOT, ".") + " " + os.environ.get("MSM_OPTION", "") + " " + os.environ.get("MSM_ATTR", "")]
for mode in os.environ.get("MSM_MODES", "f16,bf16,f32").split(","):
    model.set_precision(mode)
    if os.environ.get("MSM_TAILS_HL") and hasattr(model.sem_seg_head.predictor, "tails_hl"):
        v = os.environ["MSM_TAILS_HL"]
        model.sem_seg_head.predictor.tails_hl = {"0": False, "1": True}.get(v, tuple(v.split(",")))
    for item in filter(None, os.environ.get("MSM_ATTR", "").split(";")):       # e.g. MSM_ATTR="pixel_decoder.fpn_half_map=False"
        import ast
        path, val = item.split("=")
        obj = model.sem_seg_head
        *mods, attr = path.split(".")
        for m in mods:
            obj = getattr(obj, m)
        assert hasattr(obj, attr), path
        setattr(obj, attr, ast.literal_eval(val))
    g = model.graphed()
    for _ in range(5):
        g(feats, (480, 640))
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            g(feats, (480, 640))
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0) / 200)
    del g
    pipe = PipelinedInference(model, depth=4)
    for _ in range(4):
        pipe.submit(feats, (480, 640))
    pipe.drain()
    run = lambda: pipe.submit(None, (480, 640), slot_inputs=True)
    for _ in range(16):
        run()
    pipe.drain()
    tp = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            run()
        torch.cuda.synchronize()
        tp.append(1e3 * (time.perf_counter() - t0) / 400)
    del pipe
    out.append(f"{mode}: {min(ts):.4f} ms one batch, {min(tp):.4f} ms with four in flight ({8e3 / min(tp):.0f} images/s)")
print(" | ".join(out))
