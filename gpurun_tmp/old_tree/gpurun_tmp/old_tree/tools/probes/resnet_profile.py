"""The ResNet-50 backbone in bf16 (batch 8 at 480x640), a few eager passes -- run under rocprofv3 --kernel-trace --stats for its per-kernel table."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unseenobjectswithmeanshift_amd.resnet_backbone import ResNet50Backbone  # noqa: E402

bb = ResNet50Backbone().to("cuda").eval()
bb.backbone_dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
images = torch.randn(8, 3, 480, 640, device="cuda")
with torch.no_grad():
    for _ in range(8):
        bb(images)
torch.cuda.synchronize()
