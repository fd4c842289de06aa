"""Generate golden vectors by running the REFERENCE code (imported from /root/reference with
import-time stubs, see _ref_import.py) on seeded synthetic inputs.

Run in the build container only:   python tests/golden/make_golden.py
Outputs small ``.npz`` fixtures next to this file.  Inputs and weights are NOT stored: they are
regenerated from ``unseenobjectswithmeanshift_amd.synthetic`` by name/seed, so a fixture holds
only the reference's outputs (plus tiny explicit inputs for the smallest cases).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_import as R  # noqa: E402
from unseenobjectswithmeanshift_amd import synthetic as syn  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def packbits(b):
    return np.packbits(b.detach().cpu().numpy().astype(np.uint8).reshape(-1))


def sample_idx(numel, k=8192, seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (k,), generator=g)


# ------------------------------------------------------------------------------------------
def g_position_encoding():
    PE = R.ref("modeling.transformer_decoder.position_encoding")
    out = {}
    for n, (h, w) in [(128, (15, 20)), (32, (30, 40)), (32, (3, 2)), (128, (4, 6))]:
        pe = PE.PositionEmbeddingSine(n, normalize=True)
        out[f"pe_{n}_{h}x{w}"] = pe(torch.zeros(2, 1, h, w))
    save("position_encoding", **out)


def g_hypersphere_attention():
    AU = R.ref("modeling.transformer_decoder.attention_util")
    g = torch.Generator().manual_seed(11)
    q = torch.randn(16, 10, 32, generator=g)
    k = torch.randn(16, 37, 32, generator=g)
    v = torch.randn(16, 37, 32, generator=g)
    m = torch.rand(16, 10, 37, generator=g) < 0.4
    m[:, :, 0] = False                    # keep every row attendable
    addm = torch.zeros(16, 10, 37)
    addm[m] = float("-inf")
    o, a = AU.hypersphere_attention(q, k, v, addm)
    o2, a2 = AU.hypersphere_attention(q, k, v, None)
    # module-level: cross attention with packed weights and a bool mask
    E, H, L, S, N = 256, 8, 10, 37, 2
    shapes = {"in_proj_weight": (3 * E, E), "in_proj_bias": (3 * E,),
              "out_proj.weight": (E, E), "out_proj.bias": (E,)}
    sd = syn.synth_state_dict(shapes, salt=5)
    attn = AU.MeanShiftAttention(E, H).eval()
    attn.load_state_dict(sd, strict=True)
    query = torch.randn(L, N, E, generator=g)
    key = torch.randn(S, N, E, generator=g)
    value = torch.randn(S, N, E, generator=g)
    bm = torch.rand(N * H, L, S, generator=g) < 0.5
    bm[:, :, 3] = False
    with torch.no_grad():
        y = attn(query, key, value, attn_mask=bm)[0]
        y_nomask = attn(query, key, value)[0]
    save("hypersphere_attention", q=q, k=k, v=v, mask=m, out=o, attn=a, out_nomask=o2,
         attn_nomask=a2, query=query, key=key, value=value, bool_mask=bm, mha_out=y,
         mha_out_nomask=y_nomask)


def build_ref_decoder(dec_layers=9, dim_ff=2048, num_queries=100, salt=0):
    DEC = R.ref("modeling.transformer_decoder.meanshiftformer_transformer_decoder")
    dec = DEC.MeanShiftTransformerDecoder(
        in_channels=64, mask_classification=True, num_classes=2, hidden_dim=256,
        num_queries=num_queries, nheads=8, dim_feedforward=dim_ff, dec_layers=dec_layers,
        pre_norm=False, mask_dim=256, enforce_input_project=False,
        use_meanshift_cross_attention=True, disable_attention_mask=False,
        use_meanshift_self_attention=True, decoder_block_norm=True).eval()
    shapes = syn.decoder_param_shapes(dec_layers=dec_layers, dim_feedforward=dim_ff,
                                      num_queries=num_queries)
    ref_shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    assert ref_shapes == {k: tuple(v) for k, v in shapes.items()}, "decoder state-dict layout drifted"
    assert list(ref_shapes) == list(shapes)
    dec.load_state_dict(syn.synth_state_dict(shapes, salt=salt), strict=True)
    return dec


def g_decoder():
    # small: 64x96 image, B=2 -- everything stored
    dec = build_ref_decoder()
    x, mf = syn.synth_decoder_inputs(2, 64, 96, seed=1)
    with torch.no_grad():
        out = dec(x, mf)
    arrs = {"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"]}
    for i, a in enumerate(out["aux_outputs"]):
        arrs[f"aux{i}_logits"] = a["pred_logits"]
        arrs[f"aux{i}_masks"] = a["pred_masks"].half()
    save("decoder_small", **arrs)

    # full 480x640 shapes, B=1 -- logits + sampled mask values + packed sign bits
    x, mf = syn.synth_decoder_inputs(1, 480, 640, seed=2)
    with torch.no_grad():
        out = dec(x, mf)
    pm = out["pred_masks"]
    idx = sample_idx(pm.numel())
    arrs = {"pred_logits": out["pred_logits"], "mask_sample_idx": idx,
            "mask_sample_val": pm.flatten()[idx], "mask_sign_bits": packbits(pm > 0),
            "mask_absmax": pm.abs().max()}
    for i, a in enumerate(out["aux_outputs"]):
        arrs[f"aux{i}_logits"] = a["pred_logits"]
        arrs[f"aux{i}_sign_bits"] = packbits(a["pred_masks"] > 0)
        arrs[f"aux{i}_sample_val"] = a["pred_masks"].flatten()[idx]
    save("decoder_480x640", **arrs)


def g_msda():
    F_ = R.ref("modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    # (1) the reference's own known-answer harness (OPS/test.py:24-63): seed 3, value = rand*0.01
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    torch.manual_seed(3)
    arrs = {}
    for tag in ("double", "float"):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        aw = torch.rand(N, Lq, M, L, P) + 1e-5
        aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
        if tag == "double":
            o = F_.ms_deform_attn_core_pytorch(value.double(), shapes, loc.double(), aw.double())
        else:
            o = F_.ms_deform_attn_core_pytorch(value, shapes, loc, aw)
        arrs.update({f"t_{tag}_value": value, f"t_{tag}_loc": loc, f"t_{tag}_aw": aw, f"t_{tag}_out": o})
    # (2) realistic layout: 8 heads x 8 dims, 3 levels x 4 points, locations spilling over borders
    g = torch.Generator().manual_seed(21)
    shp = [(15, 20), (8, 10), (4, 5)]
    S = sum(h * w for h, w in shp)
    N, M, D, L, P = 2, 8, 8, 3, 4
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, S, M, L, P, 2, generator=g) * 1.3 - 0.15
    aw = torch.softmax(torch.randn(N, S, M, L * P, generator=g), -1).view(N, S, M, L, P)
    o = F_.ms_deform_attn_core_pytorch(value, torch.as_tensor(shp), loc, aw)
    arrs.update({"r_shapes": np.array(shp), "r_value": value, "r_loc": loc, "r_aw": aw, "r_out": o})
    save("msda_core", **arrs)


def g_msda_bwd():
    """Gradients of the reference's PyTorch op (functions/ms_deform_attn_func.py:52-72) by autograd in fp64 --
    the ground truth the reference's own gradcheck (OPS/test.py:66-89) measures the CUDA backward against."""
    F_ = R.ref("modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    arrs = {}

    def grads(tag, value, shapes, loc, aw, gout):
        v, l, a = (t.double().clone().requires_grad_(True) for t in (value, loc, aw))
        o = F_.ms_deform_attn_core_pytorch(v, torch.as_tensor(shapes), l, a)
        o.backward(gout.double())
        arrs.update({f"{tag}_shapes": np.array(shapes), f"{tag}_value": value, f"{tag}_loc": loc, f"{tag}_aw": aw,
                     f"{tag}_gout": gout, f"{tag}_out": o.detach(), f"{tag}_gvalue": v.grad, f"{tag}_gloc": l.grad,
                     f"{tag}_gaw": a.grad})

    # (1) OPS/test.py:66-89 recipe: N=1, M=2, Lq=2, L=2, P=2, shapes (6,4),(3,2), value = rand*0.01, seed 3
    torch.manual_seed(3)
    shapes = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes)
    for ch in (30, 32, 64):
        value = torch.rand(1, S, 2, ch, dtype=torch.float64) * 0.01
        loc = torch.rand(1, 2, 2, 2, 2, 2, dtype=torch.float64)
        aw = torch.rand(1, 2, 2, 2, 2, dtype=torch.float64) + 1e-5
        aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
        gout = torch.rand(1, 2, 2 * ch, dtype=torch.float64)
        grads(f"t{ch}", value, shapes, loc, aw, gout)
    # (2) pixel-decoder layout: 8 heads x 8 dims, 3 levels x 4 points, locations spilling over the borders
    g = torch.Generator().manual_seed(22)
    shp = [(15, 20), (8, 10), (4, 5)]
    S = sum(h * w for h, w in shp)
    N, M, D, L, P, Lq = 2, 8, 8, 3, 4, 48
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    gout = torch.randn(N, Lq, M * D, generator=g)
    grads("r", value, shp, loc, aw, gout)
    for k in ("r_out", "r_gvalue", "r_gloc", "r_gaw"):      # fp64 autograd results, stored rounded to fp32
        arrs[k] = arrs[k].float()
    save("msda_backward", **arrs)


def build_ref_pixel_decoder(salt=0):
    MSD = R.ref("modeling.pixel_decoder.msdeformattn")
    SS = R._ShapeSpec
    shape = {"res2": SS(channels=256, stride=4), "res3": SS(channels=512, stride=8),
             "res4": SS(channels=1024, stride=16), "res5": SS(channels=2048, stride=32)}
    pd = MSD.MSDeformAttnPixelDecoder(
        shape, transformer_dropout=0.0, transformer_nheads=8, transformer_dim_feedforward=1024,
        transformer_enc_layers=6, conv_dim=64, mask_dim=256, norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    shapes = syn.pixel_decoder_param_shapes()
    ref_shapes = {k: tuple(v.shape) for k, v in pd.state_dict().items()}
    assert ref_shapes == {k: tuple(v) for k, v in shapes.items()}, \
        (set(ref_shapes) ^ set(shapes), [k for k in shapes if k in ref_shapes and ref_shapes[k] != tuple(shapes[k])])
    pd.load_state_dict(syn.synth_state_dict(shapes, salt=salt), strict=True)
    return pd


def g_pixel_decoder():
    pd = build_ref_pixel_decoder()
    feats = syn.synth_backbone_features(2, 64, 96, seed=3)
    with torch.no_grad():
        mf, enc0, ms = pd.forward_features(feats)
    save("pixel_decoder_small", mask_features=mf, ms0=ms[0], ms1=ms[1], ms2=ms[2])
    feats = syn.synth_backbone_features(1, 480, 640, seed=4)
    with torch.no_grad():
        mf, enc0, ms = pd.forward_features(feats)
    idx = sample_idx(mf.numel(), k=16384)
    save("pixel_decoder_480x640", mf_sample_idx=idx, mf_sample_val=mf.flatten()[idx],
         mf_mean=mf.mean(), mf_std=mf.std(), ms0=ms[0], ms1=ms[1].half(),
         ms2_sample_val=ms[2].flatten()[idx % ms[2].numel()])


def g_mean_shift():
    MS = R.ref("modeling.transformer_decoder.mean_shift")
    arrs = {}
    # pieces on a small problem
    X, _ = syn.synth_unit_embeddings(2000, 64, clusters=6, sigma=0.15, seed=1)
    np.random.seed(3)
    first = np.random.randint(0, X.shape[0])
    np.random.seed(3)
    seeds, sel = MS.select_smart_seeds(X, 20, return_selected_indices=True)
    W = MS.ball_kernel(seeds, X, 20)
    Z = MS.seed_hill_climbing_ball(X, seeds, 20, max_iters=10)
    cc = MS.connected_components(Z, 0.04)
    # connected_components on a hand-made chain that exercises the label-mode branch
    chain = torch.nn.functional.normalize(
        torch.tensor([[1, 0, 0], [1, 0.5, 0], [1, 0.25, 0], [0, 0, 1], [1, 0.75, 0], [0, 0.1, 1]],
                     dtype=torch.float32), dim=1)
    cc_chain = MS.connected_components(chain, 0.04)
    arrs.update(s_first=first, s_sel=sel, s_seeds=seeds, s_kernel_sum=W.sum(1), s_kernel_row0=W[0, :256],
                s_Z=Z, s_cc=cc, chain=chain, cc_chain=cc_chain)
    # end to end (MS:192-229 with epsilon = 2*0.02) on planted clusters, two sizes
    for tag, n, k, S in (("a", 4800, 8, 50), ("b", 19200, 12, 100)):
        X, ids = syn.synth_unit_embeddings(n, 64, clusters=k, sigma=0.15, seed=10 + k)
        np.random.seed(3)
        first = np.random.randint(0, n)
        np.random.seed(3)
        labels, sel = MS.mean_shift_smart_init(X, kappa=20, num_seeds=S, max_iters=10)
        arrs.update({f"{tag}_first": first, f"{tag}_labels": labels.to(torch.int16), f"{tag}_sel": sel})
    # noisy variant: 2% background points (every seed tends to stay a singleton)
    X, ids = syn.synth_unit_embeddings(4800, 64, clusters=8, sigma=0.15, seed=33, background_frac=0.02)
    np.random.seed(3)
    first = np.random.randint(0, 4800)
    np.random.seed(3)
    labels, sel = MS.mean_shift_smart_init(X, kappa=20, num_seeds=50, max_iters=10)
    arrs.update(n_first=first, n_labels=labels.to(torch.int16), n_sel=sel)
    save("mean_shift", **arrs)




# ------------------------------------------------------------------------------------------
# two-stage harness (lib/fcn/test_utils.py, lib/fcn/test_dataset.py): torch/numpy-only functions
# executed from the reference sources with a namespace standing in for their module globals
# ------------------------------------------------------------------------------------------
class _Inst:
    """Just enough of detectron2.structures.Instances for get_confident_instances/combine_masks."""

    def __init__(self, **f):
        self.f = f

    def __getattr__(self, k):
        return self.__dict__["f"][k]

    def get(self, k):
        return self.f[k]

    def __getitem__(self, idx):
        return _Inst(**{k: v[idx] for k, v in self.f.items()})


def harness_inputs(seed, H=96, W=128, n_inst=7):
    """Synthetic instance predictions + depth for the harness: a few random rectangles/ellipses."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    masks = torch.zeros(n_inst, H, W)
    for i in range(n_inst):
        cy, cx = torch.rand(1, generator=g).item() * H, torch.rand(1, generator=g).item() * W
        ry, rx = 6 + torch.rand(1, generator=g).item() * 18, 6 + torch.rand(1, generator=g).item() * 24
        masks[i] = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1).float()
    scores = torch.rand(n_inst, generator=g) * 0.6 + 0.35
    classes = (torch.rand(n_inst, generator=g) < 0.8).long()
    image = torch.rand(1, 3, H, W, generator=g)
    z = 0.4 + 1.2 * torch.rand(1, 1, H, W, generator=g)
    z[torch.rand(1, 1, H, W, generator=g) < 0.3] = 0
    z[:, :, : H // 3, : W // 3] = 0                       # a region without depth
    depth = torch.cat([torch.rand(1, 2, H, W, generator=g), z], 1)
    return masks, scores, classes, image, depth


def g_harness():
    import torch.nn.functional as F_
    ns_mask = R.ref_functions("lib/utils/mask.py", ["mask_to_tight_box_numpy", "mask_to_tight_box_pytorch", "mask_to_tight_box"],
                              {"torch": torch, "np": np})
    util = type("U", (), {"mask_to_tight_box": staticmethod(ns_mask["mask_to_tight_box"])})
    cfg = type("C", (), {"device": "cpu", "TRAIN": type("T", (), {"SYN_CROP_SIZE": 224})})
    td = R.ref_functions("lib/fcn/test_dataset.py", ["crop_rois", "match_label_crop", "filter_labels_depth"],
                         {"torch": torch, "F": F_, "cfg": cfg, "util_": util, "np": np})
    nms_ns = R.ref_functions("lib/fcn/nms.py", ["nms"], {"np": np})
    tu = R.ref_functions("lib/fcn/test_utils.py", ["get_confident_instances", "combine_masks", "combine_masks_with_NMS"],
                         {"torch": torch, "np": np, "nms": nms_ns["nms"]})
    arrs = {}
    for case, seed in enumerate((1, 2, 3)):
        masks, scores, classes, image, depth = harness_inputs(seed)
        inst = _Inst(pred_masks=masks, scores=scores, pred_classes=classes)
        conf = tu["get_confident_instances"]({"instances": inst}, topk=False, score=0.6)
        conf_topk = tu["get_confident_instances"]({"instances": inst}, topk=True, low_threshold=0.4)
        label = tu["combine_masks"](conf)
        label_topk = tu["combine_masks"](conf_topk)
        bin_mask, score_mask, bbox = tu["combine_masks_with_NMS"](conf)
        out_label = torch.as_tensor(label).unsqueeze(0)
        filt = td["filter_labels_depth"](out_label, depth, 0.5)
        rgb_crops, mask_crops, rois, depth_crops = td["crop_rois"](image, filt.clone(), depth)
        # second stage stand-in: per crop a deterministic 2-instance labelling derived from the crop mask
        labels_crop = torch.zeros(rgb_crops.shape[0], 224, 224)
        for i in range(rgb_crops.shape[0]):
            m = mask_crops[i]
            labels_crop[i] = m * (2 + (torch.arange(224)[None, :] > 100).float())   # labels 2 / 3 inside the mask
            labels_crop[i][:20, :20] = 5                                               # a spurious blob (mostly outside)
        refined, labels_crop_out = td["match_label_crop"](filt, labels_crop.clone(), mask_crops, rois, depth_crops)
        refined_nodepth, _ = td["match_label_crop"](filt, labels_crop.clone(), mask_crops, rois, None)
        arrs.update({f"c{case}_label": label.astype(np.int16), f"c{case}_label_topk": label_topk.astype(np.int16),
                     f"c{case}_nms_label": bin_mask.astype(np.int16), f"c{case}_nms_score": score_mask.astype(np.int16),
                     f"c{case}_nms_bbox": bbox, f"c{case}_filt": filt.to(torch.int16),
                     f"c{case}_rois": rois, f"c{case}_rgb_crops": rgb_crops[:, :, ::3, ::3], f"c{case}_mask_crops": packbits(mask_crops > 0),
                     f"c{case}_depth_crops": depth_crops[:, :, ::3, ::3], f"c{case}_refined": refined.to(torch.int16),
                     f"c{case}_refined_nodepth": refined_nodepth.to(torch.int16),
                     f"c{case}_labels_crop_out": labels_crop_out[:, ::2, ::2].to(torch.int8)})
    save("harness", **arrs)


def g_ucn():
    """UCN / RGB-D configuration: SimpleBasePixelDecoder + PretrainedMeanShiftTransformerDecoder (one
    level, every pixel a key, attention mask at mask resolution), small full-resolution map."""
    DEC = R.ref("modeling.transformer_decoder.meanshiftformer_transformer_decoder")
    FPN = R.ref("modeling.pixel_decoder.fpn")
    SS = R._ShapeSpec
    pd = FPN.SimpleBasePixelDecoder({"res5": SS(channels=64, stride=1)}, conv_dim=64, mask_dim=256, norm="GN").eval()
    pd_shapes = {"mask_features.weight": (256, 64, 3, 3), "mask_features.bias": (256,)}
    assert {k: tuple(v.shape) for k, v in pd.state_dict().items()} == pd_shapes
    pd.load_state_dict(syn.synth_state_dict(pd_shapes, salt=3), strict=True)
    dec = DEC.PretrainedMeanShiftTransformerDecoder(
        in_channels=64, mask_classification=True, num_classes=2, hidden_dim=256, num_queries=100, nheads=8,
        dim_feedforward=2048, dec_layers=6, pre_norm=False, mask_dim=256, enforce_input_project=False,
        use_meanshift_cross_attention=True, disable_attention_mask=False, use_meanshift_self_attention=True,
        decoder_block_norm=True).eval()
    shapes = syn.decoder_param_shapes(dec_layers=6, num_feature_levels=1)
    assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
    dec.load_state_dict(syn.synth_state_dict(shapes, salt=4), strict=True)
    X, _ = syn.synth_unit_embeddings(2 * 32 * 48, 64, clusters=7, sigma=0.3, seed=21)
    feat = X.view(2, 32 * 48, 64).transpose(1, 2).reshape(2, 64, 32, 48).contiguous()     # unit-norm along C (PM:299)
    with torch.no_grad():
        mf, _, ms = pd.forward_features({"res5": feat})
        out = dec(ms, mf)
    arrs = {"mask_features": mf.half(), "pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"]}
    for i, a in enumerate(out["aux_outputs"]):
        arrs[f"aux{i}_logits"] = a["pred_logits"]
        arrs[f"aux{i}_sign_bits"] = packbits(a["pred_masks"] > 0)
    save("ucn_small", **arrs)


def g_ucn_backbone():
    """UCN RGB-D backbone: the reference's own Resnet34_8s towers (lib/networks/resnet_dilated.py, resnet.py) with the
    SEGNET glue of lib/networks/SEG.py:104-117 (add fusion, L2 normalisation) on a 64x96 frame."""
    import importlib.util
    import types
    import torch.nn.functional as F_
    pkg = types.ModuleType("refnetworks")
    pkg.__path__ = ["/root/reference/lib/networks"]
    sys.modules["refnetworks"] = pkg
    for name in ("resnet", "resnet_dilated"):
        spec = importlib.util.spec_from_file_location(f"refnetworks.{name}", f"/root/reference/lib/networks/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refnetworks.{name}"] = mod
        spec.loader.exec_module(mod)
    RD = sys.modules["refnetworks.resnet_dilated"]
    shapes = syn.ucn_backbone_param_shapes()
    sd = syn.ucn_backbone_state_dict(shapes, salt=6)
    towers = {}
    for t in ("fcn", "fcn_depth"):
        net = RD.Resnet34_8s(num_classes=64, input_channels=3, pretrained=False).eval()
        ref_shapes = {f"{t}.{k}": tuple(v.shape) for k, v in net.state_dict().items()}
        mine = {k: tuple(v) for k, v in shapes.items() if k.startswith(t + ".")}
        assert ref_shapes == mine and list(ref_shapes) == list(mine), "backbone state-dict layout drifted"
        net.load_state_dict({k[len(t) + 1:]: v for k, v in sd.items() if k.startswith(t + ".")}, strict=True)
        towers[t] = net
    g = torch.Generator().manual_seed(31)
    img = torch.randn(2, 3, 64, 96, generator=g)
    depth = torch.randn(2, 3, 64, 96, generator=g) * 0.5
    with torch.no_grad():
        rgb = towers["fcn"](img)
        feats = F_.normalize(rgb + towers["fcn_depth"](depth), p=2, dim=1)           # SEG.py:104-117
        rgb_only = F_.normalize(rgb, p=2, dim=1)                                      # INPUT 'COLOR' (SEG.py:99-100)
    # inputs are regenerated from the seed by the tests (tests/test_backbone_cpu.py::backbone_inputs)
    save("ucn_backbone", feats=feats[:, :, ::3, ::3].contiguous(), rgb_only=rgb_only[:, :, ::6, ::6].contiguous())


def _head_outputs(pd, dec, feats):
    with torch.no_grad():
        mf, _, ms = pd.forward_features(feats)
        return dec(ms, mf)


def _head_outputs_fp64(pd, dec, feats):
    """The same reference modules evaluated in float64 (parameters and inputs promoted; the pixel decoder's explicit
    ``.float()`` of its inputs, msdeformattn.py:320, is redirected to ``.double()`` for the duration).  Ten discrete attention
    masks make the head chaotic in its last digits: a mask logit within rounding of zero flips a key, and that can move a
    query by 1e-3 .. 1e-1 a few layers later.  The fp64 run is the exact value both fp32 evaluations (the reference's and the
    HIP path's) approximate; the fixtures carry it so that the tests can hold the HIP path to 'as close to exact as the
    reference's own fp32 run'."""
    import copy
    pd64, dec64 = copy.deepcopy(pd).double(), copy.deepcopy(dec).double()
    orig = torch.Tensor.float
    torch.Tensor.float = lambda self, *a, **k: self.double()
    try:
        return _head_outputs(pd64, dec64, {k: v.double() for k, v in feats.items()})
    finally:
        torch.Tensor.float = orig


def g_head_b8():
    """BASELINE configs[1] as benchmarked: batch 8 at 640x480 through the reference pixel decoder -> 9-layer decoder (the
    inputs are bench.py's own: synth_backbone_features(8, 480, 640, seed=10)).  Stored, from the reference evaluated in
    float32 AND in float64: class logits of all ten predictions of all 8 images; for EVERY image the packed sign bits of the
    final masks (the float64 bits as the XOR against the float32 bits: a sparse map), the |logit| < 2e-4 map and 8192 sampled
    values; for images 0 and 5 also the sign bits of three intermediate predictions."""
    pd, dec = build_ref_pixel_decoder(), build_ref_decoder()
    feats = syn.synth_backbone_features(8, 480, 640, seed=10)
    out = _head_outputs(pd, dec, feats)
    out64 = _head_outputs_fp64(pd, dec, feats)
    pm, pm64 = out["pred_masks"], out64["pred_masks"]
    arrs = {"pred_logits": out["pred_logits"], "mask_absmax": pm.abs().amax((1, 2, 3)), "pred_logits64": out64["pred_logits"].float()}
    idx = sample_idx(pm[0].numel())
    for b in range(8):
        b32, b64 = packbits(pm[b] > 0), packbits(pm64[b] > 0)
        arrs.update({f"b{b}_sign_bits": b32, f"b{b}_sample_val": pm[b].flatten()[idx],
                     f"b{b}_near_zero": packbits(pm[b].abs() < 2e-4),
                     f"b{b}_sign_xor64": b32 ^ b64, f"b{b}_sample_val64": pm64[b].flatten()[idx].float()})
    for b in (0, 5):
        for i in (0, 4, 8):
            a = out["aux_outputs"][i]
            a32, a64 = packbits(a["pred_masks"][b] > 0), packbits(out64["aux_outputs"][i]["pred_masks"][b] > 0)
            arrs[f"b{b}_aux{i}_sign_bits"] = a32
            arrs[f"b{b}_aux{i}_near_zero"] = packbits(a["pred_masks"][b].abs() < 2e-4)
            arrs[f"b{b}_aux{i}_sign_xor64"] = a32 ^ a64
    arrs["mask_sample_idx"] = idx
    for i, a in enumerate(out["aux_outputs"]):
        arrs[f"aux{i}_logits"] = a["pred_logits"]
        arrs[f"aux{i}_logits64"] = out64["aux_outputs"][i]["pred_logits"].float()
    save("head_480x640_b8", **arrs)


def g_head_cfg5():
    """BASELINE configs[4] hot path: 1280x960, 300 queries, 19 decoder layers (20 predictions), batch 1, through the reference
    pixel decoder -> decoder.  Stored: class logits, sampled final-mask values, packed sign bits of every 6th query."""
    pd = build_ref_pixel_decoder()
    dec = build_ref_decoder(dec_layers=19, num_queries=300)
    feats = syn.synth_backbone_features(1, 960, 1280, seed=9)
    out = _head_outputs(pd, dec, feats)
    out64 = _head_outputs_fp64(pd, dec, feats)
    pm, pm64 = out["pred_masks"][0], out64["pred_masks"][0]
    idx = sample_idx(pm.numel(), k=16384)
    qs = torch.arange(0, 300, 6)
    arrs = {"pred_logits": out["pred_logits"], "mask_sample_idx": idx, "mask_sample_val": pm.flatten()[idx], "queries": qs,
            "sign_bits": packbits(pm[qs] > 0), "near_zero": packbits(pm[qs].abs() < 2e-4), "mask_absmax": pm.abs().max(),
            "positive_fraction": (pm > 0).float().mean(),
            "pred_logits64": out64["pred_logits"].float(), "mask_sample_val64": pm64.flatten()[idx].float(),
            "sign_bits64": packbits(pm64[qs] > 0), "positive_fraction64": (pm64 > 0).float().mean()}
    for i in (0, 9, 18):
        arrs[f"aux{i}_logits"] = out["aux_outputs"][i]["pred_logits"]
        arrs[f"aux{i}_sign_bits"] = packbits(out["aux_outputs"][i]["pred_masks"][0][qs] > 0)
        arrs[f"aux{i}_near_zero"] = packbits(out["aux_outputs"][i]["pred_masks"][0][qs].abs() < 2e-4)
        arrs[f"aux{i}_logits64"] = out64["aux_outputs"][i]["pred_logits"].float()
        arrs[f"aux{i}_sign_bits64"] = packbits(out64["aux_outputs"][i]["pred_masks"][0][qs] > 0)
    save("head_cfg5_960x1280", **arrs)


def g_head_b8_seeds():
    """Three more batches of 8 at 640x480 (input seeds 11, 12, 13; same weights) through the reference pixel decoder -> 9-layer
    decoder, float32 only: what the pooled bf16 parity test needs -- class logits, for every image the packed sign bits of the
    final masks and 8192 sampled values.  With head_480x640_b8 (seed 10) that is 4 x 8 images = 3200 masks: single chaotic
    events average out, arithmetic differences remain."""
    pd, dec = build_ref_pixel_decoder(), build_ref_decoder()
    for seed in (11, 12, 13):
        feats = syn.synth_backbone_features(8, 480, 640, seed=seed)
        out = _head_outputs(pd, dec, feats)
        pm = out["pred_masks"]
        idx = sample_idx(pm[0].numel())
        arrs = {"pred_logits": out["pred_logits"], "mask_absmax": pm.abs().amax((1, 2, 3)), "mask_sample_idx": idx, "seed": seed}
        for b in range(8):
            arrs.update({f"b{b}_sign_bits": packbits(pm[b] > 0), f"b{b}_sample_val": pm[b].flatten()[idx]})
        save(f"head_480x640_b8_s{seed}", **arrs)


def g_head_cfg5_l20():
    """BASELINE configs[4] hot path as bench.py times it: 1280x960, 300 queries, **20** decoder layers (21 predictions), batch 1
    (SURVEY 8d states the config with 20 layers; head_cfg5_960x1280 holds the 19 the reference builds from DEC_LAYERS = 20).
    Same contents as g_head_cfg5."""
    pd = build_ref_pixel_decoder()
    dec = build_ref_decoder(dec_layers=20, num_queries=300)
    feats = syn.synth_backbone_features(1, 960, 1280, seed=9)
    out = _head_outputs(pd, dec, feats)
    out64 = _head_outputs_fp64(pd, dec, feats)
    pm, pm64 = out["pred_masks"][0], out64["pred_masks"][0]
    idx = sample_idx(pm.numel(), k=16384)
    qs = torch.arange(0, 300, 6)
    arrs = {"pred_logits": out["pred_logits"], "mask_sample_idx": idx, "mask_sample_val": pm.flatten()[idx], "queries": qs,
            "sign_bits": packbits(pm[qs] > 0), "near_zero": packbits(pm[qs].abs() < 2e-4), "mask_absmax": pm.abs().max(),
            "positive_fraction": (pm > 0).float().mean(),
            "pred_logits64": out64["pred_logits"].float(), "mask_sample_val64": pm64.flatten()[idx].float(),
            "sign_bits64": packbits(pm64[qs] > 0), "positive_fraction64": (pm64 > 0).float().mean()}
    for i in (0, 9, 19):
        arrs[f"aux{i}_logits"] = out["aux_outputs"][i]["pred_logits"]
        arrs[f"aux{i}_sign_bits"] = packbits(out["aux_outputs"][i]["pred_masks"][0][qs] > 0)
        arrs[f"aux{i}_near_zero"] = packbits(out["aux_outputs"][i]["pred_masks"][0][qs].abs() < 2e-4)
        arrs[f"aux{i}_logits64"] = out64["aux_outputs"][i]["pred_logits"].float()
        arrs[f"aux{i}_sign_bits64"] = packbits(out64["aux_outputs"][i]["pred_masks"][0][qs] > 0)
    save("head_cfg5_960x1280_l20", **arrs)


def g_head_cfg5_l20_frames():
    """The other three frames of the batch of FOUR that bench.py times for configs[4] (input seeds 19, 29, 39; frame 0 = seed 9 is
    head_cfg5_960x1280_l20): 1280x960, 300 queries, 20 decoder layers, through the reference pixel decoder -> decoder in float32, one frame
    at a time.  Stored per frame: class logits, 16 384 sampled final-mask values, packed sign bits and near-zero bits of every 6th query."""
    pd = build_ref_pixel_decoder()
    dec = build_ref_decoder(dec_layers=20, num_queries=300)
    qs = torch.arange(0, 300, 6)
    arrs = {"queries": qs, "seeds": torch.tensor([19, 29, 39])}
    for seed in (19, 29, 39):
        feats = syn.synth_backbone_features(1, 960, 1280, seed=seed)
        out = _head_outputs(pd, dec, feats)
        pm = out["pred_masks"][0]
        idx = sample_idx(pm.numel(), k=16384)
        arrs.update({f"s{seed}_pred_logits": out["pred_logits"], f"s{seed}_mask_sample_idx": idx, f"s{seed}_mask_sample_val": pm.flatten()[idx],
                     f"s{seed}_sign_bits": packbits(pm[qs] > 0), f"s{seed}_near_zero": packbits(pm[qs].abs() < 2e-4),
                     f"s{seed}_mask_absmax": pm.abs().max(), f"s{seed}_positive_fraction": (pm > 0).float().mean()})
    save("head_cfg5_960x1280_l20_frames", **arrs)


def g_ucn_full():
    """UCN / RGB-D configuration at full size: SimpleBasePixelDecoder + PretrainedMeanShiftTransformerDecoder over the
    480x640 embedding map -- 307 200 keys per image, attention mask at mask resolution, 6 layers, batch 1."""
    DEC = R.ref("modeling.transformer_decoder.meanshiftformer_transformer_decoder")
    FPN = R.ref("modeling.pixel_decoder.fpn")
    SS = R._ShapeSpec
    pd = FPN.SimpleBasePixelDecoder({"res5": SS(channels=64, stride=1)}, conv_dim=64, mask_dim=256, norm="GN").eval()
    pd_shapes = {"mask_features.weight": (256, 64, 3, 3), "mask_features.bias": (256,)}
    pd.load_state_dict(syn.synth_state_dict(pd_shapes, salt=3), strict=True)
    dec = DEC.PretrainedMeanShiftTransformerDecoder(
        in_channels=64, mask_classification=True, num_classes=2, hidden_dim=256, num_queries=100, nheads=8,
        dim_feedforward=2048, dec_layers=6, pre_norm=False, mask_dim=256, enforce_input_project=False,
        use_meanshift_cross_attention=True, disable_attention_mask=False, use_meanshift_self_attention=True,
        decoder_block_norm=True).eval()
    dec.load_state_dict(syn.synth_state_dict(syn.decoder_param_shapes(dec_layers=6, num_feature_levels=1), salt=4), strict=True)
    X, _ = syn.synth_unit_embeddings(480 * 640, 64, clusters=12, sigma=0.3, seed=5)
    feat = X.view(1, 480 * 640, 64).transpose(1, 2).reshape(1, 64, 480, 640).contiguous()
    with torch.no_grad():
        mf, _, ms = pd.forward_features({"res5": feat})
        out = dec(ms, mf)
    pm = out["pred_masks"][0]
    idx = sample_idx(pm.numel(), k=16384)
    qs = torch.arange(0, 100, 10)
    arrs = {"pred_logits": out["pred_logits"], "mask_sample_idx": idx, "mask_sample_val": pm.flatten()[idx], "queries": qs,
            "sign_bits": packbits(pm[qs] > 0), "near_zero": packbits(pm[qs].abs() < 2e-4), "mask_absmax": pm.abs().max(),
            "mf_sample_val": mf.flatten()[idx % mf.numel()]}
    for i, a in enumerate(out["aux_outputs"]):
        arrs[f"aux{i}_logits"] = a["pred_logits"]
    save("ucn_480x640", **arrs)


def decoder_backward_loss(out, B, Q, h, w, seed=5):
    """A fixed random linear functional of every prediction of the decoder (final + aux): the scalar whose gradient the
    decoder-backward fixture holds.  Weights are regenerated from the seed by the tests."""
    g = torch.Generator().manual_seed(seed)
    preds = out["aux_outputs"] + [{"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"]}]
    loss = 0.0
    for p in preds:
        wl = torch.randn(B, Q, p["pred_logits"].shape[-1], generator=g, dtype=torch.float64)
        wm = torch.randn(B, Q, h, w, generator=g, dtype=torch.float64) / (h * w) ** 0.5
        loss = loss + (p["pred_logits"].double() * wl.to(p["pred_logits"].device)).sum() + (p["pred_masks"].double() * wm.to(p["pred_masks"].device)).sum()
    return loss


def g_decoder_backward():
    """f3: gradients of the reference decoder (fp64 autograd through the imported MeanShiftTransformerDecoder, 64x96 frame,
    batch 2) of decoder_backward_loss with respect to its inputs and every parameter.  Stored: the loss, the input gradients
    (mask_features subsampled), per-parameter gradient norms, and the full gradient of every parameter up to 70 000 elements."""
    dec = build_ref_decoder().double()
    dec.train(False)
    x, mf = syn.synth_decoder_inputs(2, 64, 96, seed=1)
    x = [t.double().requires_grad_(True) for t in x]
    mf = mf.double().requires_grad_(True)
    out = dec(x, mf)
    loss = decoder_backward_loss(out, 2, 100, 16, 24)
    loss.backward()
    arrs = {"loss": loss.detach(), "g_x0": x[0].grad, "g_x1": x[1].grad, "g_x2": x[2].grad, "g_mf_sub": mf.grad[:, ::4].contiguous(),
            "g_mf_norm": mf.grad.norm()}
    names, norms = [], []
    for n, p in dec.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        names.append(n)
        norms.append(float(g.norm()))
        if p.numel() <= 70000:
            arrs["g__" + n] = g.float()
    arrs["param_names"] = np.array(names)
    arrs["param_grad_norms"] = np.array(norms)
    for k in list(arrs):
        if isinstance(arrs[k], torch.Tensor) and arrs[k].dtype == torch.float64 and k != "loss":
            arrs[k] = arrs[k].float()
    save("decoder_backward", **arrs)


def g_instance_inference():
    """instance_inference (pretrained_meanshiftformer_model.py:461-497) executed from the reference source with stand-ins for
    the three detectron2 containers it touches.  Pins the top-k over Q*K class scores, the class labels, the binary masks
    and the score arithmetic.  NOT pinned: BitMasks.get_bounding_boxes (detectron2 v0.6 code, absent) -- the stand-in returns
    nothing, boxes stay out of the fixture."""
    import torch.nn.functional as F_

    class Inst:
        def __init__(self, image_size):
            self.image_size = image_size

    class Boxes:
        def __init__(self, t):
            self.tensor = t

    class BitMasks:
        def __init__(self, t):
            self.tensor = t

        def get_bounding_boxes(self):
            return None

    ns = R.ref_method("MSMFormer/meanshiftformer/pretrained_meanshiftformer_model.py", "PretrainedMeanShiftMaskFormer",
                      ["instance_inference"], {"torch": torch, "F": F_, "Instances": Inst, "Boxes": Boxes, "BitMasks": BitMasks})
    arrs = {}
    for case, (Q, K, h, w, topk, seed, blobs) in enumerate([(100, 2, 30, 40, 20, 1, True), (100, 2, 120, 160, 20, 2, True),
                                                            (30, 1, 16, 24, 10, 3, False), (100, 2, 15, 20, 100, 4, True)]):
        me = type("M", (), {})()
        me.sem_seg_head = type("H", (), {"num_classes": K})()
        me.device, me.num_queries, me.test_topk_per_image, me.panoptic_on = "cpu", Q, topk, False
        mask_cls, low = syn.synth_instance_inputs(Q, h, w, num_classes=K, seed=seed, blobs=blobs)
        up = F_.interpolate(low[None], size=(4 * h, 4 * w), mode="bilinear", align_corners=False)[0]       # PM:337-343
        res = ns["instance_inference"](me, mask_cls, up)
        # which (query, class) pairs were kept: recover the query index of every kept mask by matching the thresholded maps
        scores = torch.softmax(mask_cls, -1)[:, :-1].flatten()
        kept_scores, kept = scores.topk(topk, sorted=False)
        assert torch.equal(res.pred_classes, kept % K)
        assert torch.equal(res.pred_masks, (up[kept // K] > 0).float())
        arrs.update({f"c{case}_cfg": np.array([Q, K, h, w, topk, seed, int(blobs)]), f"c{case}_pair": kept,
                     f"c{case}_classes": res.pred_classes, f"c{case}_scores": res.scores,
                     f"c{case}_mask_bits": packbits(res.pred_masks > 0), f"c{case}_mask_area": res.pred_masks.flatten(1).sum(1)})
    save("instance_inference", **arrs)


def detectron2_resnet50_keys():
    """State-dict layout of detectron2's build_resnet_backbone at DEPTH 50, FrozenBN (its default NORM;
    Base-COCO-InstanceSegmentation.yaml:2-15 leaves NORM commented out), STRIDE_IN_1X1 False: BasicStem ``stem.conv1`` and
    BottleneckBlocks ``res{2..5}.{i}.{conv1,conv2,conv3[,shortcut]}``, every Conv2d followed by a FrozenBatchNorm2d stored under
    ``<conv>.norm`` with buffers weight / bias / running_mean / running_var (no num_batches_tracked).  detectron2 is not
    installed here: written from its documented module layout, independently of resnet_backbone.py."""
    keys = {}

    def conv(name, cout, cin, k):
        keys[f"{name}.weight"] = (cout, cin, k, k)
        for b in ("weight", "bias", "running_mean", "running_var"):
            keys[f"{name}.norm.{b}"] = (cout,)

    conv("stem.conv1", 64, 3, 7)
    cin = 64
    for stage, (blocks, bott, cout) in {"res2": (3, 64, 256), "res3": (4, 128, 512), "res4": (6, 256, 1024), "res5": (3, 512, 2048)}.items():
        for i in range(blocks):
            if i == 0:
                conv(f"{stage}.{i}.shortcut", cout, cin, 1)
            conv(f"{stage}.{i}.conv1", bott, cin, 1)
            conv(f"{stage}.{i}.conv2", bott, bott, 3)
            conv(f"{stage}.{i}.conv3", cout, bott, 1)
            cin = cout
    return keys


def g_checkpoint_keys():
    """Key list + shapes (NAMES ONLY, no values) of the checkpoints the reference publishes (README.md:86-95): what
    detectron2's checkpointer saves for the meta-arch PretrainedMeanShiftMaskFormer under the two shipped configuration
    families -- ``pretrained_backbone.*`` (the attribute the meta-arch keeps its backbone under, also for the ResNet-50,
    pretrained_meanshiftformer_model.py:148-158), ``sem_seg_head.pixel_decoder.*`` / ``sem_seg_head.predictor.*`` from the
    reference's own modules built as the yamls configure them, ``criterion.empty_weight`` (SetCriterion's buffer).  pixel_mean /
    pixel_std are non-persistent buffers (:134-135) and are not saved."""
    import importlib.util
    import json
    import types
    out = {}
    pd, dec = build_ref_pixel_decoder(), build_ref_decoder()
    r50 = {f"pretrained_backbone.{k}": list(v) for k, v in detectron2_resnet50_keys().items()}
    r50.update({f"sem_seg_head.pixel_decoder.{k}": list(v.shape) for k, v in pd.state_dict().items()})
    r50.update({f"sem_seg_head.predictor.{k}": list(v.shape) for k, v in dec.state_dict().items()})
    r50["criterion.empty_weight"] = [3]
    out["mixture_ResNet50"] = r50
    # UCN RGB-D family (mixture_UCN.yaml): SEGNET towers fcn / fcn_depth (lib/networks/SEG.py:69-71,97-110), torchvision-style BatchNorm
    pkg = types.ModuleType("refnetworks")
    pkg.__path__ = ["/root/reference/lib/networks"]
    sys.modules["refnetworks"] = pkg
    for name in ("resnet", "resnet_dilated"):
        spec = importlib.util.spec_from_file_location(f"refnetworks.{name}", f"/root/reference/lib/networks/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refnetworks.{name}"] = mod
        spec.loader.exec_module(mod)
    RD = sys.modules["refnetworks.resnet_dilated"]
    DEC = R.ref("modeling.transformer_decoder.meanshiftformer_transformer_decoder")
    FPN = R.ref("modeling.pixel_decoder.fpn")
    ucn = {}
    for t in ("fcn", "fcn_depth"):
        net = RD.Resnet34_8s(num_classes=64, input_channels=3, pretrained=False)
        ucn.update({f"pretrained_backbone.{t}.{k}": list(v.shape) for k, v in net.state_dict().items()})
    upd = FPN.SimpleBasePixelDecoder({"res5": R._ShapeSpec(channels=64, stride=1)}, conv_dim=64, mask_dim=256, norm="GN")
    udec = DEC.PretrainedMeanShiftTransformerDecoder(
        in_channels=64, mask_classification=True, num_classes=2, hidden_dim=256, num_queries=100, nheads=8,
        dim_feedforward=2048, dec_layers=6, pre_norm=False, mask_dim=256, enforce_input_project=False,
        use_meanshift_cross_attention=True, disable_attention_mask=False, use_meanshift_self_attention=True,
        decoder_block_norm=True)
    ucn.update({f"sem_seg_head.pixel_decoder.{k}": list(v.shape) for k, v in upd.state_dict().items()})
    ucn.update({f"sem_seg_head.predictor.{k}": list(v.shape) for k, v in udec.state_dict().items()})
    ucn["criterion.empty_weight"] = [3]
    out["mixture_UCN"] = ucn
    path = os.path.join(HERE, "checkpoint_keys.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"checkpoint_keys: {os.path.getsize(path) / 1024:.1f} KiB, {len(r50)} + {len(ucn)} keys")


if __name__ == "__main__":
    which = sys.argv[1:] or ["pe", "attn", "decoder", "msda", "msda_bwd", "pixel", "ms", "harness", "ucn", "ucn_backbone", "inst", "head_b8", "head_cfg5", "ucn_full", "dec_bwd", "ckpt_keys", "head_b8_seeds", "head_cfg5_l20", "head_cfg5_l20_frames"]
    fns = {"pe": g_position_encoding, "attn": g_hypersphere_attention, "decoder": g_decoder,
           "msda": g_msda, "msda_bwd": g_msda_bwd, "pixel": g_pixel_decoder, "ms": g_mean_shift, "harness": g_harness, "ucn": g_ucn, "ucn_backbone": g_ucn_backbone,
           "inst": g_instance_inference, "dec_bwd": g_decoder_backward, "ckpt_keys": g_checkpoint_keys, "head_b8": g_head_b8, "head_cfg5": g_head_cfg5, "ucn_full": g_ucn_full,
           "head_b8_seeds": g_head_b8_seeds, "head_cfg5_l20": g_head_cfg5_l20, "head_cfg5_l20_frames": g_head_cfg5_l20_frames}
    for w in which:
        fns[w]()
