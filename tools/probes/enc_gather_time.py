"""msda_enc_lp gather + encoder tail of the 16-bit plans, graph-timed per entry point inside a real pass (bench.entry_graph_ms) -- for A/B
builds under $MSM_TREE."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tree = os.environ.get("MSM_TREE")
if tree:
    sys.path.insert(0, os.path.abspath(tree))
sys.path.append(ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import unseenobjectswithmeanshift_amd  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
if os.environ.get("MSM_OPTION"):                      # e.g. MSM_OPTION=ENC_NO_COOP=2
    from unseenobjectswithmeanshift_amd import _lib
    name, val = os.environ["MSM_OPTION"].split("=")
    _lib.set_option(name, int(val))
model = bench.build_model(dev)
feats = {k: v.to(dev) for k, v in syn.synth_backbone_features(8, 480, 640, seed=10).items()}
model.set_precision(os.environ.get("MSM_PRECISION", "f16"))
step = lambda: model.inference(feats, (480, 640))
step()
out = [os.path.dirname(unseenobjectswithmeanshift_amd.__file__).replace(ROOT, ".")]
for name in ("ms_deform_attn_encoder_lp", "encoder_block_hm", "conv1x1_in_multi", "dec_heads", "dec_post_self", "dec_post_cross"):
    ms, n = bench.entry_graph_ms(step, name)
    out.append(f"{name} {1e3 * ms / max(n, 1):.2f} us x {n}")
print(" | ".join(out))
