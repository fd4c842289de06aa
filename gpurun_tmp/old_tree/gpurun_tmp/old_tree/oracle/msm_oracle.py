"""CPU oracle for the MSMFormer inference hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain fp32 torch/NumPy on the CPU, the algorithms of the reference
(YoungSean/UnseenObjectsWithMeanShift) that the HIP kernels replace.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- never by
the product package, which must fail loudly when its HIP library is missing.

Parity pinning: every function below is checked against golden vectors captured from the
reference's own code imported in the build container (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``; see ``tests/test_oracle_vs_golden.py``).  Exceptions -- PARITY UNPINNED:
``instance_inference`` / ``mask_boxes`` follow pretrained_meanshiftformer_model.py:461-497 and the
documented behaviour of detectron2 v0.6 ``BitMasks.get_bounding_boxes``; detectron2 is not
installed here so the reference code for those two helpers cannot be executed.

Reference file aliases (paths relative to the reference root):
  AU  = MSMFormer/meanshiftformer/modeling/transformer_decoder/attention_util.py
  DEC = MSMFormer/meanshiftformer/modeling/transformer_decoder/meanshiftformer_transformer_decoder.py
  PE  = MSMFormer/meanshiftformer/modeling/transformer_decoder/position_encoding.py
  MSD = MSMFormer/meanshiftformer/modeling/pixel_decoder/msdeformattn.py
  OPS = MSMFormer/meanshiftformer/modeling/pixel_decoder/ops
  MS  = lib/utils/mean_shift.py
  TD  = lib/fcn/test_dataset.py
  PM  = MSMFormer/meanshiftformer/pretrained_meanshiftformer_model.py
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

KAPPA_ATTN = 30.0      # AU:26
KAPPA_CLUSTER = 20.0   # TD:51
EMBEDDING_ALPHA = 0.02  # lib/fcn/config.py:255 (epsilon = 2*alpha, MS:123)


# ----------------------------------------------------------------------------------------------
# position encoding (PE:12-52)
# ----------------------------------------------------------------------------------------------
def position_embedding_sine(batch, height, width, num_pos_feats, temperature=10000.0,
                            scale=2.0 * math.pi):
    """normalize=True variant used by both decoders (DEC:414-415, MSD:240-241).
    Returns (batch, 2*num_pos_feats, H, W): first half is the y code, second half the x code,
    channels interleave sin (even) / cos (odd) (PE:43-50)."""
    eps = 1e-6
    ys = torch.arange(1, height + 1, dtype=torch.float32)      # cumsum of ones (PE:33)
    xs = torch.arange(1, width + 1, dtype=torch.float32)
    ys = ys / (ys[-1] + eps) * scale                           # PE:37
    xs = xs / (xs[-1] + eps) * scale
    i = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / num_pos_feats)  # PE:41
    py = ys[:, None] / dim_t                                   # (H, N)
    px = xs[:, None] / dim_t                                   # (W, N)

    def interleave(p):
        out = torch.empty_like(p)
        out[:, 0::2] = p[:, 0::2].sin()
        out[:, 1::2] = p[:, 1::2].cos()
        return out

    py, px = interleave(py), interleave(px)
    pos = torch.empty(2 * num_pos_feats, height, width, dtype=torch.float32)
    pos[:num_pos_feats] = py.t()[:, :, None]
    pos[num_pos_feats:] = px.t()[:, None, :]
    return pos[None].expand(batch, -1, -1, -1).contiguous()


# ----------------------------------------------------------------------------------------------
# hypersphere (vMF) attention (AU:30-82, AU:198-432)
# ----------------------------------------------------------------------------------------------
def _unit(x, dim=-1, eps=1e-12):
    # F.normalize semantics: x / max(||x||, eps)
    return x / x.norm(dim=dim, keepdim=True).clamp_min(eps)


def hypersphere_attention(q, k, v, attn_mask=None, kappa=KAPPA_ATTN):
    """q (B,Nt,E), k/v (B,Ns,E), attn_mask additive float (B,Nt,Ns) or None.
    Returns (out (B,Nt,E), attn (B,Nt,Ns)) (AU:64-82)."""
    logits = kappa * torch.matmul(_unit(q), _unit(k).transpose(-2, -1))
    if attn_mask is not None:
        logits = logits + attn_mask
    attn = torch.softmax(logits, dim=-1)
    return _unit(torch.matmul(attn, v)), attn


def meanshift_attention(query, key, value, in_w, in_b, out_w, out_b, nheads, masked=None):
    """MeanShiftAttention.forward with the packed-weight, three-linear branch (AU:134-140).
    query (L,N,E); key/value (S,N,E); masked: bool (N*h, L, S), True = may not attend
    (AU:411-414).  Returns (L,N,E)."""
    L, N, E = query.shape
    S = key.shape[0]
    hd = E // nheads
    wq, wk, wv = in_w[:E], in_w[E:2 * E], in_w[2 * E:]
    bq, bk, bv = in_b[:E], in_b[E:2 * E], in_b[2 * E:]
    q = F.linear(query, wq, bq).reshape(L, N * nheads, hd).transpose(0, 1)   # AU:364
    k = F.linear(key, wk, bk).reshape(S, N * nheads, hd).transpose(0, 1)
    v = F.linear(value, wv, bv).reshape(S, N * nheads, hd).transpose(0, 1)
    add = None
    if masked is not None:
        add = torch.zeros(masked.shape, dtype=torch.float32)
        add[masked] = float("-inf")
    o, _ = hypersphere_attention(q, k, v, add)
    o = o.transpose(0, 1).reshape(L, N, E)                                     # AU:424
    return F.linear(o, out_w, out_b)


# ----------------------------------------------------------------------------------------------
# transformer decoder (DEC:343-695)
# ----------------------------------------------------------------------------------------------
def _ln(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def prediction_heads(sd, output, mask_features, target_size, nheads, want_mask=True):
    """DEC:660-682.  output (Q,B,E).  Returns (cls (B,Q,K+1), mask (B,Q,H,W),
    attn_mask bool (B*h, Q, h_t*w_t) or None)."""
    d = _ln(output, sd["decoder_norm.weight"], sd["decoder_norm.bias"]).transpose(0, 1)
    cls = F.linear(d, sd["class_embed.weight"], sd["class_embed.bias"])
    e = d
    for j in range(3):
        e = F.linear(e, sd[f"mask_embed.layers.{j}.weight"], sd[f"mask_embed.layers.{j}.bias"])
        if j < 2:
            e = F.relu(e)
    mask = torch.einsum("bqc,bchw->bqhw", e, mask_features)
    attn = None
    if want_mask:
        m = F.interpolate(mask, size=tuple(target_size), mode="bilinear", align_corners=False)
        m = (m.sigmoid().flatten(2) < 0.5)                                   # (B,Q,hw)
        attn = m[:, None].expand(-1, nheads, -1, -1).flatten(0, 1)          # (B*h,Q,hw) DEC:678
    return cls, mask, attn


def decoder_forward(sd, x, mask_features, nheads=8, dec_layers=9, decoder_block_norm=True,
                    num_feature_levels=3, return_trace=False):
    """MeanShiftTransformerDecoder.forward, post-norm, meanshift cross+self attention, attention
    masks enabled (DEC:540-658).  ``sd`` is the module's state dict; ``x`` the three feature maps
    (B,C,H_l,W_l); mask_features (B,mask_dim,H/4,W/4)."""
    assert len(x) == num_feature_levels
    B = x[0].shape[0]
    E = sd["query_feat.weight"].shape[1]
    src, pos, sizes = [], [], []
    for i in range(num_feature_levels):
        h, w = x[i].shape[-2:]
        sizes.append((h, w))
        pos.append(position_embedding_sine(B, h, w, E // 2).flatten(2).permute(2, 0, 1))
        if f"input_proj.{i}.weight" in sd:
            y = F.conv2d(x[i], sd[f"input_proj.{i}.weight"], sd[f"input_proj.{i}.bias"])
        else:
            y = x[i]
        y = y.flatten(2) + sd["level_embed.weight"][i][None, :, None]           # DEC:575
        src.append(y.permute(2, 0, 1))
    qpos = sd["query_embed.weight"][:, None, :].expand(-1, B, -1)
    out = sd["query_feat.weight"][:, None, :].expand(-1, B, -1)
    pred_cls, pred_mask, trace = [], [], []
    cls, mask, attn = prediction_heads(sd, out, mask_features, sizes[0], nheads)
    pred_cls.append(cls)
    pred_mask.append(mask)
    for i in range(dec_layers):
        lvl = i % num_feature_levels                                             # DEC:608
        full = attn.sum(-1) == attn.shape[-1]                                    # DEC:618
        attn = attn.clone()
        attn[full] = False
        if return_trace:
            trace.append(attn[::nheads].clone())                                 # (B,Q,hw)
        p = f"transformer_cross_attention_layers.{i}."
        t2 = meanshift_attention(out + qpos, src[lvl] + pos[lvl], src[lvl],
                                 sd[p + "meanshift_attn.in_proj_weight"], sd[p + "meanshift_attn.in_proj_bias"],
                                 sd[p + "meanshift_attn.out_proj.weight"], sd[p + "meanshift_attn.out_proj.bias"],
                                 nheads, attn)
        out = _ln(out + t2, sd[p + "norm.weight"], sd[p + "norm.bias"])          # DEC:255-257
        p = f"transformer_self_attention_layers.{i}."
        qk = out + qpos
        t2 = meanshift_attention(qk, qk, out,
                                 sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"],
                                 sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"],
                                 nheads, None)
        out = _ln(out + t2, sd[p + "norm.weight"], sd[p + "norm.bias"])          # DEC:178-179
        p = f"transformer_ffn_layers.{i}."
        t2 = F.linear(F.relu(F.linear(out, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                      sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        out = _ln(out + t2, sd[p + "norm.weight"], sd[p + "norm.bias"])          # DEC:300-304
        if decoder_block_norm:
            out = _unit(out)                                                     # DEC:637-638
        cls, mask, attn = prediction_heads(sd, out, mask_features,
                                           sizes[(i + 1) % num_feature_levels], nheads)
        pred_cls.append(cls)
        pred_mask.append(mask)
    res = {
        "pred_logits": pred_cls[-1],
        "pred_masks": pred_mask[-1],
        "aux_outputs": [{"pred_logits": a, "pred_masks": b} for a, b in zip(pred_cls[:-1], pred_mask[:-1])],
    }
    if return_trace:
        res["attn_mask_trace"] = trace
        res["queries"] = out
    return res


# ----------------------------------------------------------------------------------------------
# multi-scale deformable attention (OPS/src/cuda/ms_deform_im2col_cuda.cuh:38-89,242-304;
# OPS/functions/ms_deform_attn_func.py:52-72; OPS/modules/ms_deform_attn.py:82-125)
# ----------------------------------------------------------------------------------------------
def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights):
    """value (N,S,M,D); spatial_shapes list[(H,W)]; sampling_locations (N,Lq,M,L,P,2) in [0,1]
    (x,y); attention_weights (N,Lq,M,L,P).  Returns (N,Lq,M*D).

    Follows the CUDA kernel's arithmetic (not grid_sample): pixel coords h = y*H - 0.5,
    w = x*W - 0.5 (cuh:290-291); a point contributes only if -1 < h < H and -1 < w < W
    (cuh:293); 4-tap bilinear with zero for out-of-range corners (cuh:59-83)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    out = torch.zeros(N, Lq, M, D, dtype=value.dtype)
    start = 0
    n_idx = torch.arange(N)[:, None, None, None]
    m_idx = torch.arange(M)[None, None, :, None]
    for lid, (H, W) in enumerate([(int(h), int(w)) for h, w in spatial_shapes]):
        v = value[:, start:start + H * W]                       # (N,HW,M,D)
        start += H * W
        loc = sampling_locations[:, :, :, lid]                  # (N,Lq,M,P,2)
        wim = loc[..., 0] * W - 0.5
        him = loc[..., 1] * H - 0.5
        inside = (him > -1) & (wim > -1) & (him < H) & (wim < W)
        h0 = torch.floor(him)
        w0 = torch.floor(wim)
        lh, lw = him - h0, wim - w0
        h0, w0 = h0.long(), w0.long()
        acc = torch.zeros(N, Lq, M, P, D, dtype=value.dtype)
        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                           (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh, ww = h0 + dh, w0 + dw
            ok = inside & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
            idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1))            # (N,Lq,M,P)
            g = v[n_idx, idx, m_idx]                                       # (N,Lq,M,P,D)
            acc = acc + g * (wt * ok)[..., None]
        out = out + (acc * attention_weights[:, :, :, lid][..., None]).sum(3)
    return out.reshape(N, Lq, M * D)


def ms_deform_attn_core_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """The reference's pure-PyTorch form of the op (OPS/functions/ms_deform_attn_func.py:52-72), which its own test
    treats as ground truth (OPS/test.py:40,55): per level, the value plane of every (image, head) is sampled with
    F.grid_sample(bilinear, zeros padding, align_corners=False) at 2*loc - 1 and the samples are combined with the
    attention weights.  Same function as ms_deform_attn_core above (which follows the CUDA kernel's arithmetic) up to
    rounding; kept separately so that the reference's test can be reproduced with the comparison it makes."""
    import torch.nn.functional as F
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    out = value.new_zeros(N, M, D, Lq)
    begin = 0
    for lvl, (H, W) in enumerate([(int(h), int(w)) for h, w in spatial_shapes]):
        plane = value[:, begin:begin + H * W].permute(0, 2, 3, 1).reshape(N * M, D, H, W)     # (N*M, D, H, W)
        begin += H * W
        grid = (2.0 * sampling_locations[:, :, :, lvl] - 1.0).permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)
        taps = F.grid_sample(plane, grid, mode="bilinear", padding_mode="zeros", align_corners=False)   # (N*M, D, Lq, P)
        w = attention_weights[:, :, :, lvl].permute(0, 2, 1, 3).reshape(N * M, 1, Lq, P)
        out += (taps * w).sum(-1).view(N, M, D, Lq)
    return out.permute(0, 3, 1, 2).reshape(N, Lq, M * D).contiguous()


def ms_deform_attn_core_backward(value, spatial_shapes, sampling_locations, attention_weights, grad_output):
    """Analytic backward of ms_deform_attn_core, restating the col2im kernels: bilinear helper
    ms_deform_im2col_cuda.cuh:92-165 (grad_value scatter :128-160, grad_attn_weight = top_grad*val :164,
    grad_sampling_loc = (W*grad_w_weight, H*grad_h_weight)*top_grad*attn :165-166), point test :352.
    grad_output (N,Lq,M*D).  Returns (grad_value (N,S,M,D), grad_loc (N,Lq,M,L,P,2), grad_w (N,Lq,M,L,P))."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    go = grad_output.reshape(N, Lq, M, 1, D)
    gvalue = torch.zeros_like(value)
    gloc = torch.zeros_like(sampling_locations)
    gw = torch.zeros_like(attention_weights)
    n_idx = torch.arange(N)[:, None, None, None].expand(N, Lq, M, P)
    m_idx = torch.arange(M)[None, None, :, None].expand(N, Lq, M, P)
    start = 0
    for lid, (H, W) in enumerate([(int(h), int(w)) for h, w in spatial_shapes]):
        v = value[:, start:start + H * W]
        loc = sampling_locations[:, :, :, lid]
        aw = attention_weights[:, :, :, lid]                               # (N,Lq,M,P)
        wim = loc[..., 0] * W - 0.5
        him = loc[..., 1] * H - 0.5
        inside = (him > -1) & (wim > -1) & (him < H) & (wim < W)
        h0 = torch.floor(him)
        w0 = torch.floor(wim)
        lh, lw = him - h0, wim - w0
        hh_, hw_ = 1 - lh, 1 - lw
        h0, w0 = h0.long(), w0.long()
        tg = go * aw[..., None]                                            # top_grad_value (N,Lq,M,P,D)
        val = torch.zeros(N, Lq, M, P, D, dtype=value.dtype)
        gx = torch.zeros(N, Lq, M, P, dtype=value.dtype)
        gy = torch.zeros(N, Lq, M, P, dtype=value.dtype)
        for dh, dw, wt, cx, cy in ((0, 0, hh_ * hw_, -hh_, -hw_), (0, 1, hh_ * lw, hh_, -lw),
                                   (1, 0, lh * hw_, -lh, hw_), (1, 1, lh * lw, lh, lw)):
            hh, ww = h0 + dh, w0 + dw
            ok = inside & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
            idx = hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1)
            g = v[n_idx, idx, m_idx] * ok[..., None]                       # corner values, zero outside
            val = val + g * wt[..., None]
            gx = gx + (g * tg).sum(-1) * cx
            gy = gy + (g * tg).sum(-1) * cy
            contrib = tg * (wt * ok)[..., None]
            gvalue[:, start:start + H * W].index_put_((n_idx, idx, m_idx), contrib, accumulate=True)
        gw[:, :, :, lid] = (go * val).sum(-1) * inside
        gloc[:, :, :, lid, :, 0] = gx * W * inside
        gloc[:, :, :, lid, :, 1] = gy * H * inside
        start += H * W
    return gvalue, gloc, gw


def encoder_reference_points(spatial_shapes, batch):
    """MSD:141-153 with valid_ratios == 1: pixel centres / size, same for every level.
    Returns (batch, sum(HW), L, 2) in (x, y) order."""
    refs = []
    for (H, W) in spatial_shapes:
        ry = (torch.arange(H, dtype=torch.float32) + 0.5) / H
        rx = (torch.arange(W, dtype=torch.float32) + 0.5) / W
        gy, gx = torch.meshgrid(ry, rx, indexing="ij")
        refs.append(torch.stack((gx.reshape(-1), gy.reshape(-1)), -1))
    ref = torch.cat(refs, 0)                                       # (S,2)
    return ref[None, :, None, :].expand(batch, -1, len(spatial_shapes), -1).contiguous()


def ms_deform_attn_module(sd, prefix, query, reference_points, input_flatten, spatial_shapes,
                          nheads=8, n_points=4):
    """MSDeformAttn.forward (OPS/modules/ms_deform_attn.py:82-125), no padding mask."""
    N, Lq, C = query.shape
    L = len(spatial_shapes)
    value = F.linear(input_flatten, sd[prefix + "value_proj.weight"], sd[prefix + "value_proj.bias"])
    value = value.view(N, -1, nheads, C // nheads)
    off = F.linear(query, sd[prefix + "sampling_offsets.weight"], sd[prefix + "sampling_offsets.bias"])
    off = off.view(N, Lq, nheads, L, n_points, 2)
    aw = F.linear(query, sd[prefix + "attention_weights.weight"], sd[prefix + "attention_weights.bias"])
    aw = torch.softmax(aw.view(N, Lq, nheads, L * n_points), -1).view(N, Lq, nheads, L, n_points)
    normalizer = torch.tensor([[w, h] for h, w in spatial_shapes], dtype=torch.float32)   # (L,2) = (W,H)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = ms_deform_attn_core(value, spatial_shapes, loc, aw)
    return F.linear(out, sd[prefix + "output_proj.weight"], sd[prefix + "output_proj.bias"])


def pixel_decoder_forward(sd, features, nheads=8, enc_layers=6, n_points=4,
                          transformer_in_features=("res3", "res4", "res5"), fpn_feature="res2"):
    """MSDeformAttnPixelDecoder.forward_features (MSD:314-358) for the ResNet-50 configuration:
    transformer on res5,res4,res3 (in that order), one FPN level on res2, GroupNorm(32).
    Returns (mask_features, out[0], multi_scale_features[3])."""
    srcs, poss, shapes = [], [], []
    for idx, f in enumerate(transformer_in_features[::-1]):
        x = features[f].float()
        y = F.conv2d(x, sd[f"input_proj.{idx}.0.weight"], sd[f"input_proj.{idx}.0.bias"])
        y = F.group_norm(y, 32, sd[f"input_proj.{idx}.1.weight"], sd[f"input_proj.{idx}.1.bias"])
        C = y.shape[1]
        srcs.append(y)
        poss.append(position_embedding_sine(x.shape[0], x.shape[2], x.shape[3], C // 2))
        shapes.append((x.shape[2], x.shape[3]))
    B = srcs[0].shape[0]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)             # (B,S,C)
    pos = torch.cat([p.flatten(2).transpose(1, 2) + sd["transformer.level_embed"][l].view(1, 1, -1)
                     for l, p in enumerate(poss)], 1)                             # MSD:75
    ref = encoder_reference_points(shapes, B)
    for l in range(enc_layers):
        p = f"transformer.encoder.layers.{l}."
        a = ms_deform_attn_module(sd, p + "self_attn.", src + pos, ref, src, shapes, nheads, n_points)
        src = _ln(src + a, sd[p + "norm1.weight"], sd[p + "norm1.bias"])        # MSD:124-126
        f2 = F.linear(F.relu(F.linear(src, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                      sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        src = _ln(src + f2, sd[p + "norm2.weight"], sd[p + "norm2.bias"])        # MSD:116-118
    out, start = [], 0
    for (h, w) in shapes:
        out.append(src[:, start:start + h * w].transpose(1, 2).reshape(B, -1, h, w))
        start += h * w
    x = features[fpn_feature].float()
    lat = F.conv2d(x, sd["adapter_1.weight"], None)
    lat = F.group_norm(lat, 32, sd["adapter_1.norm.weight"], sd["adapter_1.norm.bias"])
    y = lat + F.interpolate(out[-1], size=lat.shape[-2:], mode="bilinear", align_corners=False)  # MSD:348
    y = F.conv2d(y, sd["layer_1.weight"], None, padding=1)
    y = F.relu(F.group_norm(y, 32, sd["layer_1.norm.weight"], sd["layer_1.norm.bias"]))
    out.append(y)
    mask_features = F.conv2d(out[-1], sd["mask_features.weight"], sd["mask_features.bias"])
    return mask_features, out[0], out[:3]


def simple_base_pixel_decoder_forward(sd, features, feature="res5"):
    """SimpleBasePixelDecoder.forward_features (pixel_decoder/fpn.py:261-284): identity on the embedding,
    mask_features = Conv3x3(64 -> mask_dim) with bias when mask_dim != 64."""
    y = features[feature]
    if "mask_features.weight" not in sd:
        return y, None, [y]
    return F.conv2d(y, sd["mask_features.weight"], sd["mask_features.bias"], padding=1), None, [y]


# ----------------------------------------------------------------------------------------------
# classic vMF mean shift (MS:11-229, TD:44-59) -- cosine metric only
# ----------------------------------------------------------------------------------------------
def ball_kernel(Z, X, kappa):
    return torch.exp(kappa * (Z @ X.t()))                                       # MS:26


def seed_hill_climbing_ball(X, Z, kappa, max_iters=10):
    for _ in range(max_iters):                                                  # MS:90-107
        Z = _unit(ball_kernel(Z, X, kappa) @ X, dim=1)
    return Z


def select_smart_seeds(X, num_seeds, first_index):
    """Farthest-point seeding with d = 0.5*(1 - x.s) (MS:155-187).  ``first_index`` is the value
    the reference draws with np.random.randint(0, n) (MS:155).  Returns (seeds, indices)."""
    n = X.shape[0]
    idx = torch.full((num_seeds,), -1, dtype=torch.long)
    idx[0] = first_index
    nearest = 0.5 * (1 - X @ X[first_index])
    for i in range(1, num_seeds):
        j = torch.argmax(nearest)                                               # first max on ties
        idx[i] = j
        nearest = torch.minimum(nearest, 0.5 * (1 - X @ X[j]))
    return X[idx].clone(), idx


def connected_components(Z, epsilon):
    """Sequential seed merging (MS:41-76): the i-th unlabeled seed claims every seed within
    epsilon (cosine distance); if that set already carries labels it takes their mode (smallest
    label on count ties, MS:30-38) instead of a fresh one."""
    n = Z.shape[0]
    labels = np.full(n, -1, dtype=np.int64)
    K = 0
    Zn = Z.numpy() if isinstance(Z, torch.Tensor) else Z
    for i in range(n):
        if labels[i] != -1:
            continue
        d = 0.5 * (1 - (Z @ Z[i]).numpy()) if isinstance(Z, torch.Tensor) else 0.5 * (1 - Zn @ Zn[i])
        comp = d <= epsilon
        seen = labels[comp]
        if np.unique(seen).shape[0] > 1:
            seen = seen[seen != -1]
            vals, counts = np.unique(seen, return_counts=True)
            lab = vals[np.argmax(counts)]
        else:
            lab = K
            K += 1
        labels[comp] = lab
    return torch.from_numpy(labels)


def mean_shift_smart_init(X, kappa=KAPPA_CLUSTER, num_seeds=100, max_iters=10, first_index=0,
                          epsilon=2 * EMBEDDING_ALPHA):
    """MS:192-229.  Returns (labels (n,), selected_indices (S,), Z (S,d), seed_labels (S,))."""
    seeds, sel = select_smart_seeds(X, num_seeds, first_index)
    Z = seed_hill_climbing_ball(X, seeds, kappa, max_iters)
    seed_labels = connected_components(Z, epsilon)
    closest = torch.argmin(0.5 * (1 - X @ Z.t()), dim=1)
    labels = seed_labels[closest]
    num = len(torch.unique(seed_labels))
    count = torch.bincount(labels, minlength=num)[:num]
    lmax = int(torch.argmax(count))
    if lmax != 0:                                                               # MS:217-227
        a = labels == 0
        b = labels == lmax
        labels = labels.clone()
        labels[a] = lmax
        labels[b] = 0
    return labels, sel, Z, seed_labels


def clustering_features(features, num_seeds=100, first_indices=None, kappa=KAPPA_CLUSTER,
                        max_iters=10):
    """TD:44-59.  features (B,C,H,W) unit-norm along C.  ``first_indices[j]`` replaces the
    reference's np.random.randint draw for image j."""
    B, C, H, W = features.shape
    out = torch.zeros(B, H, W)
    picked = []
    for j in range(B):
        X = features[j].reshape(C, -1).t().contiguous()
        fi = 0 if first_indices is None else int(first_indices[j])
        labels, sel, _, _ = mean_shift_smart_init(X, kappa, num_seeds, max_iters, fi)
        out[j] = labels.view(H, W).float()
        picked.append(sel)
    return out, picked


# ----------------------------------------------------------------------------------------------
# post-processing (PM:334-378, 461-497) -- PARITY UNPINNED (see module docstring)
# ----------------------------------------------------------------------------------------------
def mask_boxes(masks_bool):
    """detectron2 v0.6 BitMasks.get_bounding_boxes: [x_min, y_min, x_max+1, y_max+1], zeros for
    empty masks."""
    n = masks_bool.shape[0]
    boxes = torch.zeros(n, 4, dtype=torch.float32)
    xs = masks_bool.any(1)
    ys = masks_bool.any(2)
    for i in range(n):
        x = torch.where(xs[i])[0]
        y = torch.where(ys[i])[0]
        if len(x) > 0 and len(y) > 0:
            boxes[i] = torch.tensor([x[0], y[0], x[-1] + 1, y[-1] + 1], dtype=torch.float32)
    return boxes


def canonical_topk(scores_flat, k):
    """The reference calls topk(sorted=False) (PM:469) whose order is implementation-defined;
    the build's canonical order is score-descending with index ascending on ties."""
    order = sorted(range(scores_flat.numel()), key=lambda i: (-float(scores_flat[i]), i))[:k]
    return torch.tensor(order, dtype=torch.long)


def instance_inference(mask_cls, mask_pred_lowres, image_size, topk=20, padded_size=None):
    """One image.  mask_cls (Q,K+1); mask_pred_lowres (Q,h,w) logits; image_size (H,W); padded_size = the frame after
    ImageList.from_tensors padding (PM:275), default image_size.
    Upsample to the padded frame (PM:337-343), crop to the image (sem_seg_postprocess, PM:354-357; the second
    interpolation there is the identity when the requested output size is the image size) -> top-k over Q*K class
    scores (PM:466-474) -> binary masks, boxes, scores = class prob * mean sigmoid over the mask (PM:488-495)."""
    Q, K1 = mask_cls.shape
    K = K1 - 1
    up = F.interpolate(mask_pred_lowres[None], size=tuple(padded_size or image_size), mode="bilinear",
                       align_corners=False)[0][:, :image_size[0], :image_size[1]]
    scores = torch.softmax(mask_cls, -1)[:, :-1].flatten()
    idx = canonical_topk(scores, topk)
    s = scores[idx]
    classes = idx % K
    m = up[idx // K]
    binm = (m > 0).float()
    mscore = (m.sigmoid().flatten(1) * binm.flatten(1)).sum(1) / (binm.flatten(1).sum(1) + 1e-6)
    return {"pred_masks": binm, "pred_boxes": mask_boxes(m > 0), "scores": s * mscore,
            "pred_classes": classes, "query_index": idx // K}
