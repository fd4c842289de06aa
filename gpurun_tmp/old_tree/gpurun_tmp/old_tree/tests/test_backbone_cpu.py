"""UCN RGB-D backbone (unseenobjectswithmeanshift_amd/ucn_backbone.py) against the golden vectors produced by the
reference's own Resnet34_8s towers (tests/golden/make_golden.py::g_ucn_backbone).  The backbone is stock torch
convolutions, so this parity test runs on CPU; tests/test_gpu_modules.py repeats it on the GPU."""
import numpy as np
import torch

from unseenobjectswithmeanshift_amd import synthetic as syn
from unseenobjectswithmeanshift_amd.ucn_backbone import UCNBackbone


def backbone_inputs():
    """Same draws as make_golden.py::g_ucn_backbone."""
    g = torch.Generator().manual_seed(31)
    img = torch.randn(2, 3, 64, 96, generator=g)
    depth = torch.randn(2, 3, 64, 96, generator=g) * 0.5
    return img, depth


def make_backbone(device="cpu"):
    net = UCNBackbone(num_units=64, in_channels=3)
    shapes = syn.ucn_backbone_param_shapes()
    # the module keeps the reference's state-dict layout: SEGNET checkpoints load unchanged
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
    assert list(net.state_dict()) == list(shapes)
    net.load_state_dict(syn.ucn_backbone_state_dict(shapes, salt=6), strict=True)
    return net.to(device).eval()


def check_backbone(golden, device):
    g = golden("ucn_backbone")
    net = make_backbone(device)
    img, depth = (t.to(device) for t in backbone_inputs())
    feats = net(img, None, depth)
    assert feats.shape == (2, 64, 64, 96) and feats.is_contiguous()
    torch.testing.assert_close(feats.norm(dim=1).cpu(), torch.ones(2, 64, 96), rtol=1e-5, atol=1e-5)
    # BatchNorm folded into the convolutions: fp32 rounding differs from conv -> BN by ~1e-6 per layer
    torch.testing.assert_close(feats[:, :, ::3, ::3].cpu(), torch.from_numpy(g["feats"]), rtol=2e-4, atol=2e-5)
    rgb = net(img)                                  # colour only (INPUT 'COLOR')
    torch.testing.assert_close(rgb[:, :, ::6, ::6].cpu(), torch.from_numpy(g["rgb_only"]), rtol=2e-4, atol=2e-5)
    return net, feats


def test_backbone_vs_reference(golden):
    net, feats = check_backbone(golden, "cpu")
    # a changed BatchNorm buffer / parameter invalidates the folded weights
    with torch.no_grad():
        net.fcn.resnet34_8s.bn1.running_mean.add_(0.3)
    img, depth = backbone_inputs()
    assert not torch.allclose(net(img, None, depth), feats)
    try:
        net.train()(img, None, depth)
    except NotImplementedError:
        pass
    else:
        raise AssertionError("training mode must be rejected")
