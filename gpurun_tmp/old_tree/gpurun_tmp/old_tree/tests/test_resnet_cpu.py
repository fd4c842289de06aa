"""ResNet-50 backbone + checkpoint converter (SURVEY 8 f rank 4): structure, BatchNorm folding and key conversion on CPU.
detectron2 is absent, so there is no reference output to pin against (parity unpinned, stated in the module docstring)."""
import torch

from unseenobjectswithmeanshift_amd import checkpoint as ck
from unseenobjectswithmeanshift_amd.resnet_backbone import ResNet50Backbone


def _randomise(bb, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in bb.named_parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (2.0 / p[0].numel()) ** 0.5)
        for n, b in bb.named_buffers():
            if n.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
            elif n.endswith("weight"):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.25)
            else:
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
    return bb


def test_state_dict_layout_is_detectron2s():
    bb = ResNet50Backbone()
    sd = bb.state_dict()
    # DEPTH 50: 1 stem conv + 16 blocks x 3 convs + 4 projection shortcuts = 53 convolutions, each with 4 frozen-BN buffers
    convs = [k for k in sd if k.endswith(".weight") and not k.endswith("norm.weight")]
    assert len(convs) == 53 and len(sd) == 53 * 5
    for k, shape in {"stem.conv1.weight": (64, 3, 7, 7), "stem.conv1.norm.running_var": (64,),
                     "res2.0.shortcut.weight": (256, 64, 1, 1), "res2.0.conv1.weight": (64, 64, 1, 1),
                     "res2.0.conv2.weight": (64, 64, 3, 3), "res2.2.conv3.weight": (256, 64, 1, 1),
                     "res3.0.shortcut.weight": (512, 256, 1, 1), "res3.3.conv2.norm.bias": (128,),
                     "res4.5.conv3.weight": (1024, 256, 1, 1), "res5.0.conv1.weight": (512, 1024, 1, 1),
                     "res5.2.conv3.norm.running_mean": (2048,)}.items():
        assert tuple(sd[k].shape) == shape, k
    assert "res2.1.shortcut.weight" not in sd and "res5.0.shortcut.norm.weight" in sd
    # STRIDE_IN_1X1 False: the stride of a stage sits on the 3x3 convolution (and the shortcut) of its first block
    assert bb.res3[0].conv1.stride == (1, 1) and bb.res3[0].conv2.stride == (2, 2) and bb.res3[0].shortcut.stride == (2, 2)
    assert bb.res2[0].conv2.stride == (1, 1) and bb.res3[1].conv2.stride == (1, 1)
    assert {k: (v.channels, v.stride) for k, v in bb.output_shape().items()} == \
        {"res2": (256, 4), "res3": (512, 8), "res4": (1024, 16), "res5": (2048, 32)}


def test_folded_network_equals_definition():
    bb = _randomise(ResNet50Backbone()).double().eval()
    x = torch.randn(2, 3, 64, 96, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    ref = bb(x, folded=False)
    got = bb(x)
    assert {k: tuple(v.shape) for k, v in got.items()} == {"res2": (2, 256, 16, 24), "res3": (2, 512, 8, 12), "res4": (2, 1024, 4, 6),
                                                           "res5": (2, 2048, 2, 3)}
    for k in ref:
        assert got[k].is_contiguous()
        torch.testing.assert_close(got[k], ref[k], rtol=1e-9, atol=1e-9)
    # the folded plan follows in-place parameter updates
    with torch.no_grad():
        bb.res4[2].conv2.norm.running_mean.add_(0.3)
    torch.testing.assert_close(bb(x)["res5"], bb(x, folded=False)["res5"], rtol=1e-9, atol=1e-9)
    assert not torch.allclose(bb(x)["res5"], ref["res5"])


def test_reference_checkpoint_conversion_round_trip():
    """A checkpoint in the reference's layout ({"model": {pretrained_backbone.*, sem_seg_head.*, criterion.*}}) loads strictly."""
    from unseenobjectswithmeanshift_amd.meta_arch import build_resnet50_model
    model = build_resnet50_model()
    ref_sd = {}
    g = torch.Generator().manual_seed(3)
    for k, v in model.state_dict().items():
        k = "pretrained_backbone." + k[len("backbone."):] if k.startswith("backbone.") else k
        ref_sd["module." + k] = (torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()).numpy()
    ref_sd["module.criterion.empty_weight"] = torch.ones(3).numpy()
    ref_sd["module.pretrained_backbone.stem.conv1.norm.num_batches_tracked"] = torch.tensor(0).numpy()
    sd = ck.load_reference_checkpoint(model, {"model": ref_sd, "iteration": 17499})
    assert all(not k.startswith(("criterion", "module", "pretrained_backbone")) and not k.endswith("num_batches_tracked") for k in sd)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # v1 decoder checkpoints name the query features static_query (DEC:348-369)
    old = {k.replace("query_feat", "static_query"): v for k, v in sd.items()}
    import copy
    m2 = build_resnet50_model()
    meta = getattr(old, "_metadata", None)
    ck.load_reference_checkpoint(m2, {"model": old})
    assert torch.equal(m2.sem_seg_head.predictor.query_feat.weight, sd["sem_seg_head.predictor.query_feat.weight"])
    # a head-only model (features handed over by the caller) ignores the checkpoint's backbone
    from unseenobjectswithmeanshift_amd.meta_arch import MeanShiftMaskFormer, build_resnet50_head
    m3 = MeanShiftMaskFormer(backbone=None, sem_seg_head=build_resnet50_head(), num_queries=100)
    ck.load_reference_checkpoint(m3, {"model": ref_sd})
    import pytest
    with pytest.raises(RuntimeError, match="missing keys"):
        ck.load_reference_checkpoint(build_resnet50_model(), {"model": {k: v for k, v in ref_sd.items() if "res5.2" not in k}})
    ucn = ck.convert_ucn_state_dict({"module.fcn.resnet34_8s.conv1.weight": torch.zeros(64, 3, 7, 7), "module.fcn.resnet34_8s.bn1.num_batches_tracked": torch.tensor(1),
                                     "foo": torch.zeros(1)})
    assert list(ucn) == ["fcn.resnet34_8s.conv1.weight"]


def _fixture_checkpoint(layout, as_numpy):
    """A checkpoint file's content in the published layout (tests/golden/checkpoint_keys.json: names and shapes taken from the
    reference's own modules, make_golden.py::g_checkpoint_keys): seeded values, detectron2's {"model": ..., "iteration": ...}
    wrapper, DistributedDataParallel's ``module.`` prefix on every key."""
    g = torch.Generator().manual_seed(5)
    sd = {}
    for k, shape in layout.items():
        if k.endswith("num_batches_tracked"):
            t = torch.tensor(7, dtype=torch.int64)
        elif k.endswith("running_var"):
            t = torch.rand(shape, generator=g) + 0.5
        else:
            t = torch.randn(shape, generator=g) * 0.05
        sd["module." + k] = t.numpy() if as_numpy else t
    return {"model": sd, "iteration": 17499, "__author__": "fixture"}


def test_published_checkpoint_layouts_load_strictly(tmp_path):
    """f4: the key lists of the published checkpoints (README.md:86-95) -- the ResNet-50 / RGB family and the UCN / RGB-D
    family -- as files: ``load_reference_checkpoint(model, path, strict=True)`` consumes every model key, drops exactly the
    training-only entries, and reads the file tensors-only (no arbitrary unpickling), also when the values are numpy arrays as
    in detectron2's converted pickles."""
    import json
    import os
    from unseenobjectswithmeanshift_amd.meta_arch import build_resnet50_model, build_ucn_model
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "checkpoint_keys.json")) as f:
        layouts = json.load(f)
    for name, build, as_numpy in (("mixture_ResNet50", build_resnet50_model, False), ("mixture_UCN", build_ucn_model, True)):
        layout = layouts[name]
        model = build()
        path = tmp_path / f"{name}.pth"
        torch.save(_fixture_checkpoint(layout, as_numpy), path)
        sd = ck.load_reference_checkpoint(model, str(path), strict=True)
        mine = model.state_dict()
        assert set(sd) == set(mine), (sorted(set(sd) ^ set(mine))[:6])
        dropped = {k for k in layout if not (k.replace("pretrained_backbone.", "backbone.") in sd)}
        assert all(k.startswith("criterion.") for k in dropped), sorted(dropped)[:6]
        for k, v in mine.items():
            src = "pretrained_backbone." + k[len("backbone."):] if k.startswith("backbone.") else k
            assert tuple(v.shape) == tuple(layout[src]), k
            assert torch.equal(v, sd[k]), k
    # a file that needs a full unpickle is refused unless the caller opts in
    class Evil:
        def __reduce__(self):
            return (print, ("arbitrary code ran",))
    bad = tmp_path / "evil.pth"
    torch.save({"model": {"x": Evil()}}, bad)
    import pytest
    with pytest.raises(Exception):
        ck.load_checkpoint_file(str(bad))
