"""The RGB-D model end to end (towers -> head -> instances), batch 2 at 480x640, a few eager passes in the given plan -- run under
rocprofv3 --kernel-trace --stats for its per-kernel table (tools/probes/stats_table.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402
from unseenobjectswithmeanshift_amd.meta_arch import build_ucn_model  # noqa: E402

um = build_ucn_model()
um.backbone.load_state_dict(syn.ucn_backbone_state_dict(syn.ucn_backbone_param_shapes(), salt=6), strict=True)
um.sem_seg_head.pixel_decoder.load_state_dict(syn.synth_state_dict({"mask_features.weight": (256, 64, 3, 3), "mask_features.bias": (256,)}, salt=3))
um.sem_seg_head.predictor.load_state_dict(syn.synth_state_dict(syn.decoder_param_shapes(dec_layers=6, num_feature_levels=1), salt=4))
um = um.cuda().eval()
um.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
g = torch.Generator().manual_seed(5)
uin = {"image": torch.randn(2, 3, 480, 640, generator=g).cuda(), "depth": torch.rand(2, 3, 480, 640, generator=g).cuda()}
with torch.no_grad():
    for _ in range(8):
        um.inference_images(uin, (480, 640))
torch.cuda.synchronize()
