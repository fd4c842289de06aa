"""Seeded synthetic weights and inputs (no datasets / checkpoints exist offline).

Weights are a pure function of (parameter name, shape, salt): the golden generator loads them
into the *reference* modules, tests load the same tensors into the oracle and the HIP-backed
modules, and ``bench.py`` uses them as the random-init model.  Nothing here touches the
reference tree.  Scales follow the reference initialisers only loosely (xavier-like for
matrices, unit-ish LayerNorm) -- the point is non-degenerate activations, not training.
"""
import math
import zlib

import numpy as np
import torch


def _gen(name: str, salt: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (salt * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_param(name: str, shape, salt: int = 0) -> torch.Tensor:
    """Deterministic fp32 tensor for state-dict entry ``name``."""
    shape = tuple(int(s) for s in shape)
    g = _gen(name, salt)
    leaf = name.rsplit(".", 1)[-1]
    if "norm" in name and leaf == "weight" or (name.endswith(".1.weight") and len(shape) == 1):
        # LayerNorm / GroupNorm gains
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf in ("bias", "in_proj_bias") or len(shape) == 1:
        if "sampling_offsets" in name:
            # keep the reference's ring-shaped offset prior (ms_deform_attn.py:63-70) in spirit:
            # offsets of a few pixels so samples leave the centre cell and cross borders
            return 2.0 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)
    if "level_embed" in name or "query_" in name:
        return torch.randn(shape, generator=g)
    fan_out = shape[0]
    fan_in = int(np.prod(shape[1:]))
    std = math.sqrt(2.0 / (fan_in + fan_out))
    if "sampling_offsets" in name:
        std *= 4.0
    return std * torch.randn(shape, generator=g)


def synth_state_dict(shapes: dict, salt: int = 0) -> dict:
    """``shapes``: name -> shape.  Returns name -> tensor, in the same order."""
    return {k: synth_param(k, s, salt) for k, s in shapes.items()}


# ----------------------------------------------------------------------------------------------
# shapes of the two hot-path modules in the reference state-dict layout
# ----------------------------------------------------------------------------------------------
def decoder_param_shapes(in_channels=64, hidden_dim=256, num_queries=100, nheads=8,
                         dim_feedforward=2048, dec_layers=9, mask_dim=256, num_classes=2,
                         enforce_input_project=False, num_feature_levels=3):
    """Keys/shapes of MeanShiftTransformerDecoder.state_dict()
    (meanshiftformer_transformer_decoder.py:408-507)."""
    s = {}
    E = hidden_dim
    for i in range(dec_layers):
        p = f"transformer_self_attention_layers.{i}."
        s[p + "self_attn.in_proj_weight"] = (3 * E, E)
        s[p + "self_attn.in_proj_bias"] = (3 * E,)
        s[p + "self_attn.out_proj.weight"] = (E, E)
        s[p + "self_attn.out_proj.bias"] = (E,)
        s[p + "norm.weight"] = (E,)
        s[p + "norm.bias"] = (E,)
    for i in range(dec_layers):
        p = f"transformer_cross_attention_layers.{i}."
        s[p + "meanshift_attn.in_proj_weight"] = (3 * E, E)
        s[p + "meanshift_attn.in_proj_bias"] = (3 * E,)
        s[p + "meanshift_attn.out_proj.weight"] = (E, E)
        s[p + "meanshift_attn.out_proj.bias"] = (E,)
        s[p + "norm.weight"] = (E,)
        s[p + "norm.bias"] = (E,)
    for i in range(dec_layers):
        p = f"transformer_ffn_layers.{i}."
        s[p + "linear1.weight"] = (dim_feedforward, E)
        s[p + "linear1.bias"] = (dim_feedforward,)
        s[p + "linear2.weight"] = (E, dim_feedforward)
        s[p + "linear2.bias"] = (E,)
        s[p + "norm.weight"] = (E,)
        s[p + "norm.bias"] = (E,)
    s["decoder_norm.weight"] = (E,)
    s["decoder_norm.bias"] = (E,)
    s["query_feat.weight"] = (num_queries, E)
    s["query_embed.weight"] = (num_queries, E)
    s["level_embed.weight"] = (num_feature_levels, E)
    if in_channels != hidden_dim or enforce_input_project:
        for i in range(num_feature_levels):
            s[f"input_proj.{i}.weight"] = (E, in_channels, 1, 1)
            s[f"input_proj.{i}.bias"] = (E,)
    s["class_embed.weight"] = (num_classes + 1, E)
    s["class_embed.bias"] = (num_classes + 1,)
    dims = [E, E, E, mask_dim]
    for j in range(3):
        s[f"mask_embed.layers.{j}.weight"] = (dims[j + 1], dims[j])
        s[f"mask_embed.layers.{j}.bias"] = (dims[j + 1],)
    return s


def pixel_decoder_param_shapes(in_channels=(256, 512, 1024, 2048), conv_dim=64, mask_dim=256,
                               enc_layers=6, nheads=8, n_levels=3, n_points=4, d_ffn=1024):
    """Keys/shapes of MSDeformAttnPixelDecoder.state_dict() with res2..res5 inputs, transformer
    on res3..res5 and one extra FPN level (msdeformattn.py:197-290)."""
    s = {}
    C = conv_dim
    # input_proj is ordered low-res -> high-res (res5, res4, res3) (msdeformattn.py:209-216)
    for i, cin in enumerate(list(in_channels[1:])[::-1]):
        s[f"input_proj.{i}.0.weight"] = (C, cin, 1, 1)
        s[f"input_proj.{i}.0.bias"] = (C,)
        s[f"input_proj.{i}.1.weight"] = (C,)
        s[f"input_proj.{i}.1.bias"] = (C,)
    s["transformer.level_embed"] = (n_levels, C)
    for l in range(enc_layers):
        p = f"transformer.encoder.layers.{l}."
        s[p + "self_attn.sampling_offsets.weight"] = (nheads * n_levels * n_points * 2, C)
        s[p + "self_attn.sampling_offsets.bias"] = (nheads * n_levels * n_points * 2,)
        s[p + "self_attn.attention_weights.weight"] = (nheads * n_levels * n_points, C)
        s[p + "self_attn.attention_weights.bias"] = (nheads * n_levels * n_points,)
        s[p + "self_attn.value_proj.weight"] = (C, C)
        s[p + "self_attn.value_proj.bias"] = (C,)
        s[p + "self_attn.output_proj.weight"] = (C, C)
        s[p + "self_attn.output_proj.bias"] = (C,)
        s[p + "norm1.weight"] = (C,)
        s[p + "norm1.bias"] = (C,)
        s[p + "linear1.weight"] = (d_ffn, C)
        s[p + "linear1.bias"] = (d_ffn,)
        s[p + "linear2.weight"] = (C, d_ffn)
        s[p + "linear2.bias"] = (C,)
        s[p + "norm2.weight"] = (C,)
        s[p + "norm2.bias"] = (C,)
    s["mask_features.weight"] = (mask_dim, C, 1, 1)
    s["mask_features.bias"] = (mask_dim,)
    # one FPN level on res2 (norm="GN" => conv bias absent, msdeformattn.py:264-279)
    s["adapter_1.weight"] = (C, in_channels[0], 1, 1)
    s["adapter_1.norm.weight"] = (C,)
    s["adapter_1.norm.bias"] = (C,)
    s["layer_1.weight"] = (C, C, 3, 3)
    s["layer_1.norm.weight"] = (C,)
    s["layer_1.norm.bias"] = (C,)
    return s


# ----------------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------------
def synth_backbone_features(batch, height, width, in_channels=(256, 512, 1024, 2048), seed=0,
                            planted_objects=12):
    """Synthetic ResNet-50 feature pyramid res2..res5 for an ``height x width`` image
    (strides 4/8/16/32).  Post-ReLU-like (non-negative) activations with ``planted_objects``
    spatial blobs per image so that attention masks are neither all-set nor all-clear
    (SURVEY.md section 8(d) "structured" input)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)
    feats = {}
    # blob layout at stride 4, shared by all levels through area pooling
    h4, w4 = height // 4, width // 4
    yy, xx = torch.meshgrid(torch.arange(h4, dtype=torch.float32), torch.arange(w4, dtype=torch.float32),
                            indexing="ij")
    obj = torch.zeros(batch, planted_objects, h4, w4)
    for b in range(batch):
        for k in range(planted_objects):
            cy = torch.rand(1, generator=g).item() * h4
            cx = torch.rand(1, generator=g).item() * w4
            r = (0.04 + 0.08 * torch.rand(1, generator=g).item()) * min(h4, w4)
            obj[b, k] = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).float()
    for name, c, stride in zip(("res2", "res3", "res4", "res5"), in_channels, (4, 8, 16, 32)):
        h, w = height // stride, width // stride
        proto = torch.randn(planted_objects, c, generator=g)
        o = torch.nn.functional.adaptive_avg_pool2d(obj, (h, w))              # (B,K,h,w)
        x = torch.einsum("bkhw,kc->bchw", o, proto) + 0.5 * torch.randn(batch, c, h, w, generator=g)
        feats[name] = torch.relu(x).contiguous()
    return feats


def synth_decoder_inputs(batch, height, width, in_channels=64, mask_dim=256, seed=0):
    """Inputs of MeanShiftTransformerDecoder.forward: three maps at strides 32/16/8 and
    mask_features at stride 4 (meanshiftformer_transformer_decoder.py:540)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(2000 + seed)
    x = [torch.randn(batch, in_channels, height // s, width // s, generator=g) for s in (32, 16, 8)]
    mf = torch.randn(batch, mask_dim, height // 4, width // 4, generator=g) * 0.5
    return x, mf


def synth_unit_embeddings(n, d=64, clusters=12, sigma=0.15, seed=0, background_frac=0.0):
    """Planted vMF-like clusters on the unit sphere: normalize(mu_k + sigma*N(0,I)/sqrt(d))
    (SURVEY.md section 8(d) mean-shift recipe).  Returns (X (n,d) fp32 unit rows, ids (n,))."""
    g = torch.Generator(device="cpu")
    g.manual_seed(3000 + seed)
    mu = torch.nn.functional.normalize(torch.randn(clusters, d, generator=g), dim=1)
    ids = torch.randint(0, clusters, (n,), generator=g)
    X = mu[ids] + sigma * torch.randn(n, d, generator=g) / math.sqrt(d)
    if background_frac > 0:
        nb = int(n * background_frac)
        idx = torch.randperm(n, generator=g)[:nb]
        X[idx] = torch.randn(nb, d, generator=g)
        ids[idx] = -1
    X = torch.nn.functional.normalize(X, dim=1)
    return X.contiguous(), ids


# ----------------------------------------------------------------------------------------------
# UCN backbone (lib/networks/SEG.py SEGNET 'seg_resnet34_8s_embedding', RGB-D add fusion)
# ----------------------------------------------------------------------------------------------
def ucn_backbone_param_shapes(num_units=64, in_channels=3, use_depth=True):
    """Keys/shapes of the two Resnet34_8s towers in the reference's state-dict order (resnet.py:139-177,
    resnet_dilated.py:296-305)."""
    def bn(p, c, s):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)
        s[p + ".running_mean"] = (c,)
        s[p + ".running_var"] = (c,)
        s[p + ".num_batches_tracked"] = ()

    s = {}
    for tower in ("fcn", "fcn_depth")[:2 if use_depth else 1]:
        r = tower + ".resnet34_8s."
        s[r + "conv1.weight"] = (64, in_channels, 7, 7)
        bn(r + "bn1", 64, s)
        cin = 64
        for i, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
            for j in range(blocks):
                b = f"{r}layer{i + 1}.{j}."
                s[b + "conv1.weight"] = (planes, cin if j == 0 else planes, 3, 3)
                bn(b + "bn1", planes, s)
                s[b + "conv2.weight"] = (planes, planes, 3, 3)
                bn(b + "bn2", planes, s)
                if j == 0 and i > 0:
                    s[b + "downsample.0.weight"] = (planes, cin, 1, 1)
                    bn(b + "downsample.1", planes, s)
            cin = planes
        s[r + "fc.weight"] = (num_units, 512, 1, 1)
        s[r + "fc.bias"] = (num_units,)
    return s


def ucn_backbone_state_dict(shapes=None, salt=0):
    """Non-degenerate seeded weights for the backbone: He-scaled convolutions, BatchNorm gains around 1, running
    variances in [0.8, 1.2]."""
    shapes = ucn_backbone_param_shapes() if shapes is None else shapes
    out = {}
    for k, shp in shapes.items():
        g = _gen(k, salt)
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros((), dtype=torch.long)
        elif leaf == "running_var":
            out[k] = 0.8 + 0.4 * torch.rand(shp, generator=g)
        elif leaf == "running_mean":
            out[k] = 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            is_gain = leaf == "weight"
            out[k] = (1.0 if is_gain else 0.0) + (0.1 if is_gain else 0.05) * torch.randn(shp, generator=g)
        else:
            out[k] = math.sqrt(2.0 / int(np.prod(shp[1:]))) * torch.randn(shp, generator=g)
    return out


def synth_instance_inputs(num_queries=100, h=30, w=40, num_classes=2, seed=0, blobs=True):
    """Inputs of the instance post-processing (PM:337-343, 461-497) for one image: class logits (Q, K+1) and low-resolution
    mask logits (Q, h, w).  ``blobs``: smooth object-like maps (a few gaussian bumps minus an offset) so that the masks have
    interiors, borders and -- for some queries -- no positive pixel at all; else white noise."""
    g = torch.Generator().manual_seed(seed)
    mask_cls = torch.randn(num_queries, num_classes + 1, generator=g) * 2.0
    if not blobs:
        return mask_cls, torch.randn(num_queries, h, w, generator=g) * 3.0
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    m = torch.zeros(num_queries, h, w)
    for q in range(num_queries):
        for _ in range(1 + q % 3):
            cy, cx = torch.rand(1, generator=g).item() * h, torch.rand(1, generator=g).item() * w
            sy, sx = 1.5 + torch.rand(1, generator=g).item() * h / 4, 1.5 + torch.rand(1, generator=g).item() * w / 4
            m[q] += 8.0 * torch.exp(-0.5 * (((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
        m[q] -= 3.0 + 6.0 * (q % 7 == 0)                 # every seventh query: (almost) nothing above zero
    return mask_cls, m + 0.1 * torch.randn(num_queries, h, w, generator=g)

class StandInBackbone(torch.nn.Module):
    """Stand-in for a ResNet-50 feature pyramid where only the SHAPES of res2..res5 matter (two-stage harness tests and the
    configs[3] bench line, whose unit of work is the head, not the backbone): average-pool pyramid + fixed random 1x1
    mixing, plain torch ops.  res2..res5 with 256/512/1024/2048 channels for any H, W divisible by 32."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.mix = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(c, 6, generator=g) * 0.5) for c in (256, 512, 1024, 2048)])

    def forward(self, images, depth=None):
        x = images if depth is None else torch.cat([images, depth], 1)
        if x.shape[1] == 3:
            x = torch.cat([x, x], 1)
        out = {}
        p, prev = x, 1
        for name, s, w in zip(("res2", "res3", "res4", "res5"), (4, 8, 16, 32), self.mix):
            # the pyramid level by pooling the previous level (the full-resolution input is read once, not four times) and the 1x1
            # mixing as ONE bmm whose (B, C, HW) result IS the NCHW tensor (einsum / broadcasting matmul compute the transposed product
            # and copy it: 27 copy launches of 83 us per two-stage batch under rocprofv3)
            p = torch.nn.functional.avg_pool2d(p, s // prev)
            prev = s
            b, c, h, ww = p.shape
            out[name] = torch.bmm(w.unsqueeze(0).expand(b, -1, -1), p.reshape(b, c, h * ww)).relu_().view(b, -1, h, ww)
        return out
