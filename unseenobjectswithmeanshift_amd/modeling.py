"""Host-side mirror of the reference's hot-path modules, running on the HIP kernels.

Same class names, constructor keyword arguments, ``forward`` signatures, return structures and
``state_dict()`` keys/shapes as the reference (so reference checkpoints load with strict=True):

  PositionEmbeddingSine            <- transformer_decoder/position_encoding.py:12-52
  MeanShiftAttention               <- transformer_decoder/attention_util.py:434-540
  hypersphere_attention            <- transformer_decoder/attention_util.py:30-82
  MeanShiftTransformerDecoder      <- transformer_decoder/meanshiftformer_transformer_decoder.py:343-695
  MSDeformAttn                     <- pixel_decoder/ops/modules/ms_deform_attn.py:34-125
  MSDeformAttnPixelDecoder         <- pixel_decoder/msdeformattn.py:164-358

Inference only (no autograd through the kernels).  torch supplies parameters, device memory and
streams; all arithmetic is in libmsm_hip.so.  Internally tokens are batch-major (B, L, E); the
reference's (L, B, E) layout appears only at the MeanShiftAttention API boundary.
"""
import math

import torch
from torch import nn

from . import ops
from ._plan import PlanAttributes, TensorList, version_key

KAPPA = 30  # attention_util.py:26


class ShapeSpec:
    """Stand-in for detectron2.layers.ShapeSpec (channels / stride of a backbone feature)."""

    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


# ----------------------------------------------------------------------------------------------
class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        if not normalize:
            raise NotImplementedError("the hot path only uses normalize=True (DEC:415, MSD:241)")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("padding masks are not used by the MSMFormer configs")
        B, _, H, W = x.shape
        pe = ops.pos_embed_sine(H, W, self.num_pos_feats, x.device, temperature=float(self.temperature),
                                scale=float(self.scale))
        return pe[None].expand(B, -1, -1, -1)


# ----------------------------------------------------------------------------------------------
def hypersphere_attention(q, k, v, attn_mask=None, dropout_p=0.0, kappa=KAPPA):
    """Single-head form of attention_util.py:30-82 for (B, Nt, E=32*h) tensors is provided through
    MeanShiftAttention; this functional entry point takes the reference's per-head layout
    q (B*h, Nt, 32), k/v (B*h, Ns, 32) and a float mask with -inf entries, and returns the attended
    values only (the (B*h, Nt, Ns) weights are never materialised)."""
    if dropout_p > 0.0:
        raise NotImplementedError("inference only")
    Bh, Nt, E = q.shape
    if E != 32:
        raise RuntimeError("head_dim must be 32")
    masked = row_any = None
    if attn_mask is not None:
        masked = (attn_mask == float("-inf")).to(torch.uint8).contiguous()
        row_any = torch.ones((Bh, Nt), device=q.device, dtype=torch.int32)
    return ops.hypersphere_attention(q.contiguous(), k.contiguous(), v.contiguous(), 1, masked=masked,
                                     row_any=row_any, kappa=float(kappa))


class FoldedMaskFeatures:
    """The mask features of MSDeformAttnPixelDecoder in factored form: mask_features = weight . act + bias with ``act`` the
    64-channel FPN activation relu(GroupNorm(layer_1 conv)) as NCHW planes (B, 64, H, W) (msdeformattn.py:349-358).

    Every consumer of mask_features on the inference path is the bilinear contraction einsum("bqc,bchw->bqhw", e,
    mask_features) (DEC:668), and that is linear in mask_features:
        einsum(e, W a + b) = einsum(e W, a) + e.b
    so a decoder that understands this object contracts the 64-channel ``act`` with the folded embedding e W (64 columns)
    plus a per-query constant e.b -- a quarter of the FLOPs and of the bytes of the mask step, and the 1x1 convolution that
    would write the (B, 256, H, W) tensor is never run.  ``tensor()`` materialises the literal mask_features for any other
    consumer (same kernels as the unfolded pixel decoder)."""

    def __init__(self, act, weight, bias, materialize):
        self.act, self.weight, self.bias = act, weight, bias
        self._materialize = materialize
        self._tensor = None

    @property
    def shape(self):
        B, _, H, W = self.act.shape
        return torch.Size((B, self.weight.shape[0], H, W))

    @property
    def device(self):
        return self.act.device

    def tensor(self):
        if self._tensor is None:
            self._tensor = self._materialize()
        return self._tensor


class ConvFoldedMaskFeatures:
    """The mask features of SimpleBasePixelDecoder in factored form: mask_features = Conv3x3(x) + bias with ``x`` the 64-channel
    level feature (B, 64, H, W) itself (fpn.py:238-246,283-290).  The decoder's only use of mask_features on the inference path is
    einsum("bqc,bchw->bqhw", e, mask_features) (DEC:1012-1035), linear in x:
        einsum(e, W * x + b) = (e W) * x + e.b        ((e W)[q] : a 3x3 filter over 64 channels per query)
    so a decoder that understands this object convolves x with per-query filters (ops.mask_conv3x3_folded, K = 576 on the fp16
    tokens the fused K/V attention reads) and the (B, 256, H, W) tensor -- 629 MB at batch 2 of 480x640 -- is never written.
    ``tensor()`` materialises the literal mask_features for any other consumer."""

    def __init__(self, x, weight, bias, materialize):
        self.x, self.weight, self.bias = x, weight, bias
        self._materialize = materialize
        self._tensor = None

    @property
    def shape(self):
        B, _, H, W = self.x.shape
        return torch.Size((B, self.weight.shape[0], H, W))

    @property
    def device(self):
        return self.x.device

    def tensor(self):
        if self._tensor is None:
            self._tensor = self._materialize()
        return self._tensor


class MeanShiftAttention(nn.Module):
    """Parameters laid out as nn.MultiheadAttention(embed_dim, num_heads) (attention_util.py:469-472):
    in_proj_weight (3E,E), in_proj_bias (3E), out_proj.{weight,bias}."""

    def __init__(self, embed_dim, num_heads=1, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False,
                 kdim=None, vdim=None, batch_first=False, device=None, dtype=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.batch_first = False
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    # batch-major core used by the decoder: everything (B, L, E)
    def attend(self, tgt, memory_k, memory_v, *, query_pos=None, key_pos=None, masked=None, row_any=None,
               kv=None):
        E = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        q = ops.gemm(tgt, w[:E], b[:E], a2=query_pos)
        if kv is None:
            k = ops.gemm(memory_k, w[E:2 * E], b[E:2 * E], a2=key_pos)
            v = ops.gemm(memory_v, w[2 * E:], b[2 * E:])
        else:
            k, v = kv
        o = ops.hypersphere_attention(q, k, v, self.num_heads, masked=masked, row_any=row_any, kappa=float(KAPPA))
        return ops.gemm(o, self.out_proj.weight, self.out_proj.bias)

    def project_kv(self, memory, pos):
        E = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        return ops.gemm(memory, w[E:2 * E], b[E:2 * E], a2=pos), ops.gemm(memory, w[2 * E:], b[2 * E:])

    @torch.no_grad()
    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None):
        """Reference signature (attention_util.py:474-540): (L,N,E) / (S,N,E) inputs, bool attn_mask
        (N*h, L, S) -- must be identical across the heads of one batch element, as the decoder
        builds it (DEC:678).  Returns (attn_output (L,N,E), None): averaged weights are never built."""
        if key_padding_mask is not None:
            raise NotImplementedError("key_padding_mask is unused by the MSMFormer decoder (DEC:622)")
        L, N, E = query.shape
        S = key.shape[0]
        q = query.transpose(0, 1).contiguous()
        k = key.transpose(0, 1).contiguous()
        v = value.transpose(0, 1).contiguous()
        masked = row_any = None
        if attn_mask is not None:
            if attn_mask.dtype != torch.bool:
                raise NotImplementedError("only bool attention masks")
            m = attn_mask.view(N, self.num_heads, L, S)
            masked = m[:, 0].to(torch.uint8).contiguous()
            row_any = torch.ones((N, L), device=query.device, dtype=torch.int32)
        out = self.attend(q, k, v, masked=masked, row_any=row_any)
        return out.transpose(0, 1), None


# ----------------------------------------------------------------------------------------------
class MeanShiftCrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead=1, dropout=0.0, activation="relu", layer_normalize_before=False):
        super().__init__()
        if layer_normalize_before:
            raise NotImplementedError("PRE_NORM=False in every MSMFormer config")
        self.meanshift_attn = MeanShiftAttention(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)


class MeanShiftSelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        if normalize_before:
            raise NotImplementedError("PRE_NORM=False in every MSMFormer config")
        self.self_attn = MeanShiftAttention(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = ops.gemm(x, layer.weight, layer.bias, act="relu" if i < self.num_layers - 1 else None)
        return x


class MeanShiftTransformerDecoder(PlanAttributes, nn.Module):
    """meanshiftformer_transformer_decoder.py:343-695.  Only the configuration every MSMFormer yaml
    selects is implemented (post-norm, mean-shift cross + self attention, attention masks on):
    anything else raises at construction.

    ``aux_outputs``: inference consumes only the last prediction (pretrained_meanshiftformer_model.py:
    335-345), so by default the 9 intermediate (B,Q,H,W) masks are computed in registers for the
    attention-mask bits but never written; set ``self.aux_outputs = True`` to get the reference's
    full list.  ``self.sparse_taps = True`` additionally skips mask rows that feed no 2x2 tap."""

    _version = 2
    NUM_FEATURE_LEVELS = 3        # "we always use 3 scales" (DEC:494); the pretrained/UCN variant uses 1 (DEC:848)

    def __init__(self, in_channels, mask_classification=True, *, num_classes, hidden_dim, num_queries, nheads,
                 dim_feedforward, dec_layers, pre_norm, mask_dim, enforce_input_project,
                 use_meanshift_cross_attention=True, disable_attention_mask=False,
                 use_meanshift_self_attention=True, decoder_block_norm=True):
        super().__init__()
        assert mask_classification, "Only support mask classification model"
        if pre_norm or not use_meanshift_cross_attention or not use_meanshift_self_attention or disable_attention_mask:
            raise NotImplementedError("only PRE_NORM=False with mean-shift cross/self attention and attention masks")
        if hidden_dim % nheads or hidden_dim // nheads != 32:
            raise NotImplementedError("head_dim must be 32 (HIDDEN_DIM 256 / NHEADS 8)")
        self.mask_classification = mask_classification
        self.num_heads = nheads
        self.num_layers = dec_layers
        self.num_queries = num_queries
        self.decoder_block_norm = decoder_block_norm
        self.num_feature_levels = self.NUM_FEATURE_LEVELS
        self.aux_outputs = False
        self.sparse_taps = False
        # inference: the nine intermediate mask steps at the resolution of their attention masks (interpolation and contraction
        # commute: csrc/attn_mask.hip); False: every step at full resolution with the taps pooled afterwards; "always": also
        # when aux_outputs asks for every full-resolution mask (then the full-resolution kernel only writes the masks)
        self.pooled_attention_masks = True
        # inference entry of the meta-architecture: K > 0 -> the final mask step runs only for the K queries instance_inference
        # keeps (top-K class scores, PM:461-497); the output dict then holds pred_masks (B, K, H, W) and "topk" = (scores, classes,
        # query index).  0: all queries (the reference's head output)
        self.pe_layer = PositionEmbeddingSine(hidden_dim // 2, normalize=True)
        self.transformer_self_attention_layers = nn.ModuleList(
            MeanShiftSelfAttentionLayer(hidden_dim, nheads) for _ in range(dec_layers))
        self.transformer_cross_attention_layers = nn.ModuleList(
            MeanShiftCrossAttentionLayer(hidden_dim, nheads) for _ in range(dec_layers))
        self.transformer_ffn_layers = nn.ModuleList(
            FFNLayer(hidden_dim, dim_feedforward) for _ in range(dec_layers))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.query_feat = nn.Embedding(num_queries, hidden_dim)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.level_embed = nn.Embedding(self.num_feature_levels, hidden_dim)
        self.input_proj = nn.ModuleList()
        for _ in range(self.num_feature_levels):
            if in_channels != hidden_dim or enforce_input_project:
                self.input_proj.append(nn.Conv2d(in_channels, hidden_dim, kernel_size=1))
            else:
                self.input_proj.append(nn.Sequential())
        self.class_embed = nn.Linear(hidden_dim, num_classes + 1)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        self._pos_cache = {}
        self._kv_cache = None
        # K/V of every cross-attention layer come from ONE K=64 GEMM on the raw level features: input_proj,
        # level embedding, position code and the in-projection are folded into per-layer constants
        # (see _folded_kv).  False: materialise src = input_proj(x)+level_embed and project it (K=256).
        self.fold_kv = True
        # the K/V projections depend only on the level features, not on the query chain: batched_kv computes those of all
        # layers in ONE launch before the layer loop (nine launches, the coarse ones latency bound: 211 us per step at
        # B=8; one launch: see DESIGN.md).  (Running them on a side stream next to the query chain was neutral.)
        self.batched_kv = True
        # row + column tables instead of a per-position matrix for the folded K/V constants (see _folded_kv)
        self.separable_kv_constants = True
        # mask_features handed over as FoldedMaskFeatures are contracted in their 64-channel factored form (fused tails only)
        # 16-bit plans, attention masks at key resolution: "x3" (default, round 6) = both operands as hi + lo IEEE-half pairs, three terms per
        # product (fp32-class logits -- the mask bits feed back into the attention -- at six K = 32 MFMAs per key block instead of the fp32
        # chain's sixteen: 12 -> 7 us at 4800 keys); True = single IEEE-half operands (round 5: "mask step only in 16 bits" then loses its
        # 0.3 % bound on one image of eight); False = the fp32 MFMA chain
        self.lp_pooled_masks = "x3"
        self.folded_mask_features = True
        self._fold_cache = None
        # the row-local ops between the attention cores run as three fused kernels per layer (csrc/dec_chain.hip)
        # instead of 13 launches; needs E = 256, mask_dim = 256 and dim_feedforward % 256 == 0 (every MSMFormer yaml)
        self.fused_tails = (hidden_dim == 256 and mask_dim == 256 and dim_feedforward % 256 == 0)
        self._tails_cache = None
        # "bf16": the Q x pixel-embedding mask step runs with bf16 operands / fp32 accumulation on a packed copy of
        # mask_features made once per forward (BASELINE configs 3 and 5); everything else stays fp32
        self.mask_step_dtype = "f32"        # "bf16": bf16 operands; "f16": IEEE-half operands (same kernel, fp16 MFMAs); "f32_split": fp32-accurate three-term bf16 splits (folded form)
        # "bf16": the fused row-local tails (dec_post_cross / dec_post_self / dec_heads) stream bf16 weights and multiply on
        # bf16 MFMAs with fp32 accumulation (activations as hi + lo pairs); part of set_precision("bf16")
        self.tails_dtype = "f32"
        # tails_dtype "bf16": which tail launches multiply by hi + lo bf16 WEIGHT fragments (ops.dec_pack_weight_bf16x2; round 6): True = all
        # three, False = none, or a tuple of "post_cross" / "post_self" / "heads".  Measured over 3200 masks against the fp32 reference
        # (test_config2_slices_low_precision_vs_reference; single fragments everywhere: 1.08 % of the final mask bits, mean IoU 0.9513 --
        # SURVEY 8c asks 0.95): the heads' weights (the mask-embedding MLP and the next query projection) alone 0.96 % / 0.9568 for 1 % of the
        # pass; post_cross alone 1.05 % / 0.9525; post_self (the FFN, two thirds of the bytes) alone 1.07 % / 0.9517 for 6 %; all three
        # 0.92 % / 0.9583 for 8 %.  The embedding that is thresholded against every pixel is where a weight's 2^-9 shows.
        self.tails_hl = ("heads",)
        self.ffn_parts = None          # hidden-dimension slices of the fused FFN tail (None: ops.dec_post_self's default)
        # 16-bit plans with attention masks at key resolution: the next layer's mask as the epilogue of the heads kernel
        # (ops.dec_heads_mask: one launch instead of dec_heads + attn_mask_pooled; the same values bit for bit).  Opt-in: measured on
        # MI355X at B = 8, 640x480 it is NOT faster -- 14.6 / 17 / 27 us per fused launch at 300 / 1200 / 4800 keys against 11.2 + 5.2 / 6.5 /
        # 12.0 us for the pair, 1.455 against 1.448 ms per graph-replayed pass (f16), and no better with IEEE-half mask operands: the
        # contraction runs on the ONE CU that owns the 16-query tile (or on a few more that each repeat the MLP chain), where the
        # separate launch spreads it over the chip, and a kernel boundary inside a HIP graph costs ~1.5 us
        self.fused_head_masks = False
        # L2 prefetch of the fused tails' packed weights by an extra row of workgroups in the tail launch in front (ops.dec_set_prefetch,
        # csrc/dec_chain.hip PfRanges): True = in the 16-bit plans (a layer's tail weights are 3.2 MB there), "always" = in every plan, False = off
        self.weight_prefetch = True
        # "bf16": the attention cores multiply on bf16 MFMAs (fp32 accumulation, exp, sums) and the batched K/V projection
        # stores bf16; part of set_precision("bf16")
        self.attention_dtype = "f32"
        # low-precision attention only (head.set_precision("f16")): "f16" = the K columns of the K/V projection stored as IEEE half
        # and q^ / k^ on fp16 MFMAs (kappa = 30 multiplies the cosine's rounding error), "bf16" = bf16 everywhere
        self.attention_keys = "bf16"
        # 16-bit plans: the folded K/V projection of a LONG level (>= 16 384 keys, separable constants, W % 16 == 0: the 307 200-key UCN
        # path, the 120 x 160 level of configs[4]) inside the attention kernel (csrc/attention.hip, hs_attn_fkv_kernel): K / V are never
        # written.  False: msm_kv_project_multi_bf16 + msm_hypersphere_attn_lp_fwd (the tested alternative)
        self.fused_kv_attention = True
        self.fused_kv_min_keys = 4096       # levels with at least this many keys take the fused kernel (the 60 x 80 level of the headline shapes: 1337 -> 1264 us per pass)
        # True: the batched K/V projection computes its fp32 products as exact three-term bf16 splits (set_precision("f32_split"))
        self.kv_split = False
        self._packed_mf = None

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # v1 -> v2 key migration, as the reference (DEC:348-369)
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for k in list(state_dict.keys()):
                if k.startswith(prefix) and "static_query" in k:
                    state_dict[k.replace("static_query", "query_feat")] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        """cfg: any object with the reference's yacs attribute paths (DEC:510-538)."""
        mf, sh = cfg.MODEL.MASK_FORMER, cfg.MODEL.SEM_SEG_HEAD
        assert mf.DEC_LAYERS >= 1
        return dict(in_channels=in_channels, mask_classification=mask_classification,
                    num_classes=sh.NUM_CLASSES, hidden_dim=mf.HIDDEN_DIM, num_queries=mf.NUM_OBJECT_QUERIES,
                    nheads=mf.NHEADS, dim_feedforward=mf.DIM_FEEDFORWARD, dec_layers=mf.DEC_LAYERS - 1,
                    pre_norm=mf.PRE_NORM, enforce_input_project=mf.ENFORCE_INPUT_PROJ, mask_dim=sh.MASK_DIM,
                    use_meanshift_cross_attention=mf.USE_MEANSHIFT_CROSS_ATTENTION,
                    disable_attention_mask=mf.DISABLE_MEANSHIFT_ATTENTION_MASK,
                    use_meanshift_self_attention=mf.USE_MEANSHIFT_SELF_ATTENTION,
                    decoder_block_norm=mf.DECODER_BLOCK_NORM)

    def _pos_tokens(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = ops.pos_embed_sine(h, w, self.pe_layer.num_pos_feats, device, layout="tokens",
                                                      temperature=float(self.pe_layer.temperature),
                                                      scale=float(self.pe_layer.scale))
        return self._pos_cache[key]

    def _folded_kv(self, sizes, device):
        """Per layer i (level l = i % 3):  K_i = (input_proj_l(x) + level_embed_l + pos_l) Wk_i^T + bk_i
                                        V_i = (input_proj_l(x) + level_embed_l)         Wv_i^T + bv_i
        (DEC:575, AU:134-140) are affine in x, so they equal x [Wk_i Wp_l ; Wv_i Wp_l]^T + C_i with an
        input-independent (H_l W_l, 2E) matrix C_i.  Returns (weights, constants): constants[i] = (tensor, width) --
        width 0: the dense matrix; width W: the separable form ops.kv_project takes as ``cmat_width``.  Folding cuts the projection FLOPs 4x (K = 64 instead
        of 256) and removes the src tensors; the constants are evaluated in fp64 once per checkpoint/shape."""
        E = self.query_feat.weight.shape[1]
        params = [self.level_embed.weight] + [p for m in self.input_proj for p in m.parameters()] + \
                 [p for l in self.transformer_cross_attention_layers for p in (l.meanshift_attn.in_proj_weight, l.meanshift_attn.in_proj_bias)]
        pkey = tuple((p.data_ptr(), p._version) for p in params)
        # separable constants (below) everywhere but on the small maps of the bf16 mode: its K/V kernel is issue-bound and the second
        # table costs it 16 more loads per 16-token unit (measured at B = 8, 640x480 levels: 59 us dense, 69 separable; fp32 MFMA
        # kernel: 135 dense, 128 separable); from 128x128 keys on the dense matrix is the larger cost in every mode
        sep_min = (min(16384, int(self.fused_kv_min_keys)) if self.fused_kv_attention else 16384) if self.attention_dtype == "bf16" else 0
        skey = (tuple(sizes), str(device), sep_min)
        # one entry per input geometry (the two-stage harness alternates between the frame and the 224x224 crops); a
        # parameter change drops them all
        if self._kv_cache is None or self._kv_cache.get("params") != pkey:
            self._kv_cache = {"params": pkey}
        if len(self._kv_cache) > 48:                                          # bounded: params + geometries (+ their fused-K/V packs, dense copies)
            self._kv_cache = {"params": pkey}
        if skey not in self._kv_cache:
            ws, cs = [], []
            for i, layer in enumerate(self.transformer_cross_attention_layers):
                l = i % self.num_feature_levels
                h, w = sizes[l]
                a = layer.meanshift_attn
                wk, wv = a.in_proj_weight[E:2 * E].double(), a.in_proj_weight[2 * E:].double()
                bk, bv = a.in_proj_bias[E:2 * E].double(), a.in_proj_bias[2 * E:].double()
                lvl = self.level_embed.weight[l].double()
                if isinstance(self.input_proj[l], nn.Conv2d):
                    wp = self.input_proj[l].weight.view(E, -1).double()
                    off = self.input_proj[l].bias.double() + lvl
                else:
                    wp = torch.eye(E, dtype=torch.float64, device=device)
                    off = lvl
                pos = self._pos_tokens(h, w, device).double()
                ws.append(torch.cat([wk @ wp, wv @ wp], 0).float().contiguous())
                # The sine embedding is cat(pos_y, pos_x) (position_encoding.py:44-51): its first half depends on the row only,
                # its second on the column only, so the constant of token (y, x) is row[y] + col[x] -- two tables of h + w
                # vectors instead of h w (629 MB at 480x640: as many bytes as the projection writes).  Checked on the values,
                # not assumed: any other embedding keeps the dense matrix.
                pg, E2 = pos.view(h, w, E), E // 2
                if (self.separable_kv_constants and h > 1 and w > 1 and h * w >= sep_min and torch.equal(pg[:, :1, :E2].expand(h, w, E2), pg[..., :E2])
                        and torch.equal(pg[:1, :, E2:].expand(h, w, E - E2), pg[..., E2:])):
                    row = torch.cat([pg[:, 0, :E2] @ wk[:, :E2].t() + (off @ wk.t() + bk), (off @ wv.t() + bv).expand(h, -1)], 1)
                    col = torch.cat([pg[0, :, E2:] @ wk[:, E2:].t(), torch.zeros(w, E, dtype=torch.float64, device=device)], 1)
                    cs.append((torch.cat([row, col], 0).float().contiguous(), w))        # (h + w, 2E), width
                else:
                    kc = (pos + off) @ wk.t() + bk                               # (hw, E)
                    vc = (off @ wv.t() + bv).expand(h * w, -1)                   # (hw, E)
                    cs.append((torch.cat([kc, vc], 1).float().contiguous(), 0))
            # (a batched launch takes all its jobs separable or none: forward() densifies a mixed job list, see _uniform_constants)
            self._kv_cache[skey] = (ws, cs)
        return self._kv_cache[skey]

    def _heads(self, d, mask_features, target_size, want_mask, want_cls):
        cls = ops.gemm(d, self.class_embed.weight, self.class_embed.bias) if want_cls else None
        e = self.mask_embed(d)
        mask, attn, row_any = ops.mask_logits(e, mask_features, want_mask=want_mask, target_size=target_size,
                                              sparse=self.sparse_taps, packed_bf16=self._packed_mf,
                                              packed_split=getattr(self, "_packed_mf_split", None))
        return cls, mask, attn, row_any

    def _packed_tails(self):
        """Weights of the fused tails in the kernels' fragment order (ops.dec_pack_weight), re-packed when a
        parameter is replaced or modified in place."""
        E = self.query_feat.weight.shape[1]
        groups = {
            "cross_q": [l.meanshift_attn.in_proj_weight[:E] for l in self.transformer_cross_attention_layers],
            "cross_o": [l.meanshift_attn.out_proj.weight for l in self.transformer_cross_attention_layers],
            "self_in": [l.self_attn.in_proj_weight for l in self.transformer_self_attention_layers],
            "self_o": [l.self_attn.out_proj.weight for l in self.transformer_self_attention_layers],
            "ffn1": [l.linear1.weight for l in self.transformer_ffn_layers],
            "ffn2": [l.linear2.weight for l in self.transformer_ffn_layers],
            "mlp": [l.weight for l in self.mask_embed.layers],
        }
        kernel_of = {"cross_q": "heads", "mlp": "heads", "cross_o": "post_cross", "self_in": "post_cross", "self_o": "post_self", "ffn1": "post_self",
                     "ffn2": "post_self"}                     # the launch that multiplies by the group (a launch takes one fragment form)
        key = (self.tails_dtype, str(self.tails_hl)) + tuple((p.data_ptr(), p._version) for ws in groups.values() for p in ws)
        if self._tails_cache is None or self._tails_cache[0] != key:
            self._tails_cache = (key, {k: [self._tails_pack(kernel_of[k])(w.contiguous()) for w in ws] for k, ws in groups.items()})
        return self._tails_cache[1]

    def _tails_hl_for(self, kernel):
        """Whether the tail launch ``kernel`` multiplies by hi + lo bf16 weight fragments (``tails_hl``; the bf16 plan only)."""
        return self.tails_dtype == "bf16" and (self.tails_hl is True or (isinstance(self.tails_hl, (tuple, list, set, frozenset)) and kernel in self.tails_hl))

    def _tails_pack(self, kernel="heads"):
        """The packer of the weight matrices of the tail launch ``kernel`` ("post_cross", "post_self", "heads") for ``tails_dtype``: fp32
        fragments, bf16 fragments (activations as hi + lo bf16 pairs; the weights too where ``tails_hl`` names the launch) or IEEE-half
        fragments (precision "f16": one fp16 activation term, csrc/dec_chain.hip)."""
        hl = self._tails_hl_for(kernel)
        try:
            return {"f32": ops.dec_pack_weight, "bf16": ops.dec_pack_weight_bf16x2 if hl else ops.dec_pack_weight_bf16,
                    "f16": ops.dec_pack_weight_f16}[self.tails_dtype]
        except KeyError:
            raise ValueError("tails_dtype must be 'f32', 'bf16' or 'f16'") from None

    def _folded_head(self, fm):
        """Last mask_embed layer with the mask_features projection folded in (see FoldedMaskFeatures): for
        e = d2 W3^T + b3 the step needs e Wm (64 columns) and e.bm (one), i.e. a Linear with weight [Wm^T W3 ; bm^T W3]
        and bias [Wm^T b3 ; bm.b3] -- evaluated in fp64 once per parameter version, zero-padded to the 256 rows the
        heads kernel writes, packed like the other tail weights.  Returns (packed weight, bias, n_columns)."""
        l3 = self.mask_embed.layers[-1]
        params = (l3.weight, l3.bias, fm.weight) + ((fm.bias,) if fm.bias is not None else ())
        pack = self._tails_pack()
        key = (self.tails_dtype, str(self.tails_hl)) + tuple((p.data_ptr(), p._version) for p in params)
        if self._fold_cache is None or self._fold_cache[0] != key:
            wm = fm.weight.detach().double().reshape(fm.weight.shape[0], -1)             # (mask_dim, 64)
            bm = fm.bias.detach().double() if fm.bias is not None else torch.zeros(wm.shape[0], dtype=torch.float64, device=wm.device)
            w3, b3 = l3.weight.detach().double(), l3.bias.detach().double()
            n = wm.shape[1]
            w = torch.zeros_like(w3)
            b = torch.zeros_like(b3)
            w[:n] = wm.t() @ w3
            w[n] = bm @ w3
            b[:n] = wm.t() @ b3
            b[n] = bm @ b3
            self._fold_cache = (key, pack(w.float().contiguous()), b.float().contiguous(), n)
        return self._fold_cache[1:]

    @staticmethod
    def _poolable_sizes(act, sizes):
        """Level sizes that are integer reductions (2, 4, 8) of the mask-feature map: their attention masks can be computed at key
        resolution (csrc/attn_mask.hip)."""
        Hm, Wm = int(act.shape[2]), int(act.shape[3])
        out = []
        for (th, tw) in sizes:
            if (int(th), int(tw)) not in out and Hm % th == 0 and Wm % tw == 0 and Hm // th == Wm // tw and Hm // th in (2, 4, 8):
                out.append((int(th), int(tw)))
        return out

    def _keys_f16(self):
        """Whether the 16-bit plan's K rows are IEEE-half bit patterns (kv_format 2): ONE decision for the projection that writes them and
        the attention kernels that read them.  The half-key form exists for 2E = 512 only; any other width keeps bf16 keys."""
        return self.attention_dtype == "bf16" and self.attention_keys == "f16" and 2 * self.query_feat.weight.shape[1] == 512

    def _kv_one(self, x, w, cc):
        """One layer's folded K/V projection when the layers' K/V are not all resident at once (the 307 200-key UCN path): the
        plan's precision applies as in the batched form -- bf16 output from bf16 MFMAs in the bf16 mode, exact three-term splits
        under f32_split (a one-job launch of the batched kernel), the fp32 MFMA kernel otherwise."""
        c, cw = cc
        if x.shape[1] == 64 and w.shape[0] in (256, 512):
            if self.attention_dtype == "bf16":
                return ops.kv_project_multi([x], [w], [c], out_dtype=torch.bfloat16, cmat_widths=[cw],
                                            keys_f16=self._keys_f16())[0]
            if self.kv_split:
                return ops.kv_project_multi([x], [w], [c], split=True, cmat_widths=[cw])[0]
        return ops.kv_project(x, w, c, cw)

    def _uniform_constants(self, kv_c, jobs):
        """The constants of the layers ``jobs`` for ONE batched projection launch, which takes all its jobs separable or none: a mixed
        list (long levels separable, short ones dense) has its separable members expanded to the dense matrix (cached per layer)."""
        cs = [kv_c[i] for i in jobs]
        if len({cw > 0 for _, cw in cs}) <= 1:
            return cs
        out = []
        for i, (c, cw) in zip(jobs, cs):
            if cw > 0:
                key = ("dense", i, c.data_ptr(), tuple(c.shape))
                if key not in self._kv_cache:
                    self._kv_cache[key] = ops.dense_kv_constant(c, cw)
                c, cw = self._kv_cache[key], 0
            out.append((c, cw))
        return out

    def _fused_kv_plan(self, xs, sizes, kv_w, kv_c):
        """Which cross-attention layers take the fused K/V + attention kernel (see ``fused_kv_attention``), with their packed weights /
        transposed V constants (cached with the folded constants) and the fp16 token form of each such level (made once per forward).
        -> None, or {"layers": [None | (w_packed, rowcol, col_v_t)], "x": {level: (B, S, 64) float16}}."""
        if not (self.fused_kv_attention and self.fused_tails and self.attention_dtype == "bf16"):
            return None
        E = self.query_feat.weight.shape[1]
        H = self.num_heads
        cache = self._kv_cache.setdefault(("fkv", tuple(sizes), str(xs[0].device)), {})
        layers, xh = [], {}
        for i in range(self.num_layers):
            l = i % self.num_feature_levels
            h, w = sizes[l]
            c, cw = kv_c[i]
            if not (cw == w and w % 16 == 0 and h * w >= int(self.fused_kv_min_keys) and xs[l].shape[1] == 64 and tuple(kv_w[i].shape) == (2 * E, 64) and E == H * 32
                    and h * w * 128 < (1 << 32)):
                layers.append(None)
                continue
            if i not in cache:
                cache[i] = (ops.attn_pack_kv_weights(kv_w[i], H), c, c[h:, E:].t().contiguous())
            layers.append(cache[i])
            if l not in xh:
                xh[l] = ops.tokens_f16(xs[l])
        return {"layers": layers, "x": xh} if xh else None

    def _forward_fused(self, xs, sizes, kv_w, kv_c, mask_features, out, qpos, kv_all=None, final_topk=0, fkv=None):
        """Same arithmetic as the loop in forward(), with the row-local ops of a layer in three launches:
        heads (+ next cross-attention query) -> mask step -> K/V GEMM -> cross attention -> post_cross (out_proj, LN,
        self-attention in-projection) -> self attention -> post_self (out_proj, LN, FFN by hidden chunk)."""
        E = self.query_feat.weight.shape[1]
        L = self.num_layers
        full = self.aux_outputs
        H = self.num_heads
        pk = self._packed_tails()
        mlp = [(pk["mlp"][j], l.bias) for j, l in enumerate(self.mask_embed.layers)]
        ncol = None
        pooled = {}
        ra0 = None
        ra_all, fuse_masks = None, False
        fm_params = None
        if isinstance(mask_features, FoldedMaskFeatures):
            fm_params = [t for t in (mask_features.weight, mask_features.bias) if t is not None]
            # the heads kernel emits [e Wm | e.bm | 0...] instead of e; the mask step runs on the 64-channel activation
            wf, bf, ncol = self._folded_head(mask_features)
            mlp[-1] = (wf, bf)
            mask_features = mask_features.act
            if self.pooled_attention_masks and (not full or self.pooled_attention_masks == "always") and ncol == 64 and mask_features.shape[1] == 64:
                # attention masks at key resolution: pool the activation once to every level size that is an integer reduction
                want_sizes = self._poolable_sizes(mask_features, sizes)
                if want_sizes and L > 0:
                    # (the pooling launch also clears the row flags of prediction 0's attention-mask step)
                    # (... and, for the fused heads + mask launches, of every later prediction's: one (L + 1, B, Q) buffer)
                    fuse_masks = bool(self.fused_head_masks) and self.tails_dtype in ("bf16", "f16") and not self._tails_hl_for("heads") and not full \
                        and self.lp_pooled_masks != "x3"          # (the epilogue form has the fp32 and the single-half contraction)
                    Bq, Qn = int(out.shape[0]), int(out.shape[1])
                    outs, flags = ops.pool_mask_taps(mask_features, want_sizes, zero_rows=Qn * (L + 1 if fuse_masks else 1))
                    ra_all = flags.view(-1)[:Bq * Qn * (L + 1 if fuse_masks else 1)].view(-1, Bq, Qn)
                    ra0 = ra_all[0]
                    pooled = dict(zip(want_sizes, outs))
        dn = self.decoder_norm
        pred_cls, pred_mask = [], []
        topk_out = []
        cf = None
        if isinstance(mask_features, ConvFoldedMaskFeatures):
            # per-query 3x3 filters F = e W (one small GEMM per prediction) convolved with the fp16 tokens of the level: bits for the next
            # layer's fused K/V attention, fp32 logits for the final prediction (on the K kept queries when the caller selects first)
            wkey = ("cfw", str(mask_features.device)) + version_key([mask_features.weight, mask_features.bias])
            wc = getattr(self, "_conv_fold_cache", None)
            if wc is None or wc[0] != wkey:
                self._conv_fold_cache = wc = (wkey, ops.mask_conv_fold_weight(mask_features.weight, mask_features.bias))
            cf = (fkv["x"][0], wc[1])

        def predict_conv_folded(d, e, ra, i_next):
            last = i_next == L
            x16, wf = cf
            if not last:
                attn, row_any = ops.mask_conv3x3_folded(x16, ops.gemm(e, wf), sizes[0], bits=True, row_any=ra)
                pred_cls.append(None)
                pred_mask.append(None)
                return attn, row_any
            cls = ops.gemm(d, self.class_embed.weight, self.class_embed.bias)
            if 0 < final_topk < e.shape[1]:
                *topk, sel = ops.topk_class_scores(cls, int(final_topk), gather=e, gather_cols=e.shape[2])
                topk_out.append(tuple(topk))
                e = sel
            pred_cls.append(cls)
            pred_mask.append(ops.mask_conv3x3_folded(x16, ops.gemm(e, wf), sizes[0], bits=False))
            return None, None

        def predict(d, e, ra, i_next):
            if cf is not None:
                return predict_conv_folded(d, e, ra, i_next)
            last = i_next == L
            want = full or last
            cls = ops.gemm(d, self.class_embed.weight, self.class_embed.bias) if want else None
            tgt = None if (last and not full) else sizes[i_next % self.num_feature_levels]
            emb, qb = (e, None) if ncol is None else (e[..., :ncol], e[..., ncol])
            if last and not full and ncol is not None and 0 < final_topk < e.shape[1] and cls is not None:
                # only the masks instance_inference keeps: top-K class scores first, then the mask step on those K embeddings
                # (the top-K launch also copies the kept rows: (B, K, 68) = [e Wm | e.bm | pad], rows 16-byte aligned)
                *topk, sel = ops.topk_class_scores(cls, int(final_topk), gather=e, gather_cols=ncol + 4)
                topk = tuple(topk)
                # (fp32 kernel in every precision mode: one launch on K queries does not pay for a packed copy of the activation)
                m = ops.mask_logits(sel[..., :ncol], mask_features, want_mask=True, target_size=None, qbias=sel[..., ncol])[0]
                pred_cls.append(cls)
                pred_mask.append(m)
                topk_out.append(topk)
                return None, None
            if tgt is not None and tuple(tgt) in pooled:
                # (a layer whose cross-attention projects K / V itself reads its mask bit-packed: written that way here)
                as_bits = fkv is not None and i_next < L and fkv["layers"][i_next] is not None
                attn, row_any = ops.attn_mask_pooled(emb, pooled[tuple(tgt)], qbias=qb, row_any=ra, bits=as_bits,
                                                          f16=self.lp_pooled_masks if self.mask_step_dtype in ("bf16", "f16") else False)
                m = None
                if want:        # "always" with aux outputs: the full-resolution kernel only writes the mask
                    m = ops.mask_logits(emb, mask_features, want_mask=True, target_size=None, packed_bf16=self._packed_mf, qbias=qb,
                                        packed_split=getattr(self, "_packed_mf_split", None))[0]
                pred_cls.append(cls)
                pred_mask.append(m)
                return attn, row_any
            m, attn, row_any = ops.mask_logits(emb, mask_features, want_mask=want, target_size=tgt, sparse=self.sparse_taps,
                                               row_any=ra,       # ra: cleared by the heads kernel, no fill launch
                                               packed_bf16=self._packed_mf, qbias=qb, packed_split=getattr(self, "_packed_mf_split", None))
            pred_cls.append(cls)
            pred_mask.append(m)
            return attn, row_any

        def next_query(i):
            if i >= L:
                return dict(wq=None, bq=None, query_pos=None)
            return dict(wq=pk["cross_q"][i], bq=self.transformer_cross_attention_layers[i].meanshift_attn.in_proj_bias[:E],
                        query_pos=qpos)

        if not full and L > 0 and fm_params is not None:
            # prediction 0 starts from the learned queries: decoder_norm, the mask-embedding MLP and the first cross-attention query
            # do not depend on the input -- computed once per parameter version (the attention mask they feed does: it contracts
            # e0 with this pass's pooled activation)
            if getattr(self, "_heads0_params", None) is None:
                self._heads0_params = TensorList(self.parameters)
            hkey = (tuple(out.shape), str(out.device), self.tails_dtype, str(self.tails_hl)) + version_key(self._heads0_params()) + version_key(fm_params)
            hc = getattr(self, "_heads0_cache", None)
            if hc is None or hc[0] != hkey:
                _, d, e, q, _ = ops.dec_heads(out, dn.weight, dn.bias, mlp, want_out=False, want_d=False, zero_row_any=True, **next_query(0))
                self._heads0_cache = hc = (hkey, d, e, q)
            _, d, e, q = hc
            ra = ra0                                        # cleared by the pooling launch (None: the mask step clears its own)
        else:
            _, d, e, q, ra = ops.dec_heads(out, dn.weight, dn.bias, mlp, want_out=False, want_d=full or L == 0,
                                           zero_row_any=True, **next_query(0))
        attn, row_any = predict(d, e, ra, 0)
        # L2 prefetch of the tails' weights (weight_prefetch; ops.dec_set_prefetch): every tail launch carries a row of workgroups that
        # touch the packed weights of the launches BEHIND it in the chain, so those start on L2 hits instead of HBM latency
        pf_on = bool(self.weight_prefetch) and out.is_cuda and (self.weight_prefetch == "always" or self.tails_dtype in ("bf16", "f16"))

        def prefetch(tensors):
            if pf_on:
                ops.dec_set_prefetch(tensors)

        if pf_on:
            ops.dec_set_prefetch([])          # a request is consumed by the NEXT tail launch of this thread: drop one an aborted forward left behind

        for i in range(L):
            lvl = i % self.num_feature_levels                                     # DEC:608
            ca = self.transformer_cross_attention_layers[i]
            sa = self.transformer_self_attention_layers[i]
            ff = self.transformer_ffn_layers[i]
            lp = self.attention_dtype == "bf16"
            kf = self._keys_f16()
            if fkv is not None and fkv["layers"][i] is not None:
                # K / V of this level are projected inside the attention kernel and never stored
                o = ops.hypersphere_attention_fused_kv(q, fkv["x"][lvl], *fkv["layers"][i], sizes[lvl], H, masked=attn, row_any=row_any,
                                                       kappa=float(KAPPA), keys_f16=kf)
            else:
                kv = kv_all[i] if kv_all is not None else self._kv_one(xs[lvl], kv_w[i], kv_c[i])          # (B, hw, 2E) = [K | V]
                o = ops.hypersphere_attention(q, kv[..., :E], kv[..., E:], H, masked=attn, row_any=row_any, kappa=float(KAPPA), low_precision=lp,
                                              keys_f16=kf)
            prefetch([pk["self_o"][i], pk["ffn1"][i], pk["ffn2"][i]])              # post_self's weights, from post_cross's launch
            x, qk, v = ops.dec_post_cross(o, out, qpos, pk["cross_o"][i], ca.meanshift_attn.out_proj.bias, ca.norm.weight,
                                          ca.norm.bias, pk["self_in"][i], sa.self_attn.in_proj_bias)
            o = ops.hypersphere_attention(qk[..., :E], qk[..., E:], v, H, kappa=float(KAPPA), low_precision=lp, keys_f16=kf)
            # (the heads' weights -- the shared mask-embedding MLP, the next query projection -- are L2 residents already: prefetching them from
            # post_self's launch measured 1.386 against 1.380 ms per pass)
            x, parts = ops.dec_post_self(o, x, pk["self_o"][i], sa.self_attn.out_proj.bias, sa.norm.weight, sa.norm.bias,
                                         pk["ffn1"][i], ff.linear1.bias, pk["ffn2"][i], n_parts=self.ffn_parts)
            last = i == L - 1
            if not last:
                prefetch([pk["cross_o"][i + 1], pk["self_in"][i + 1]])           # the next layer's post_cross weights, from the heads' launch
            tgt = None if last else tuple(sizes[(i + 1) % self.num_feature_levels])
            if fuse_masks and cf is None and ncol is not None and tgt in pooled:
                # prediction i + 1 of a plan that keeps only the final masks: its one product is the next layer's attention mask
                as_bits = fkv is not None and fkv["layers"][i + 1] is not None
                out, d, e, q, attn, row_any = ops.dec_heads_mask(x, dn.weight, dn.bias, mlp, pooled[tgt], ra_all[i + 1], qcol=ncol, bits=as_bits,
                                                                 f16=self.mask_step_dtype in ("bf16", "f16") and self.lp_pooled_masks is True,
                                                                 parts=parts, bias=ff.linear2.bias, ln_g=ff.norm.weight, ln_b=ff.norm.bias,
                                                                 l2norm=self.decoder_block_norm, want_out=True, want_d=False, **next_query(i + 1))
                pred_cls.append(None)
                pred_mask.append(None)
                continue
            out, d, e, q, ra = ops.dec_heads(x, dn.weight, dn.bias, mlp, parts=parts, bias=ff.linear2.bias,
                                             ln_g=ff.norm.weight, ln_b=ff.norm.bias, l2norm=self.decoder_block_norm,
                                             want_out=not last, want_d=full or last, zero_row_any=True,
                                             **next_query(i + 1))
            attn, row_any = predict(d, e, ra, i + 1)
        res = {"pred_logits": pred_cls[-1], "pred_masks": pred_mask[-1], "aux_outputs": []}
        if topk_out:
            res["topk"] = topk_out[0]
        if full:
            res["aux_outputs"] = [{"pred_logits": a, "pred_masks": b} for a, b in zip(pred_cls[:-1], pred_mask[:-1])]
        return res

    def _initial_queries(self, B, dev):
        """query_feat broadcast over the batch (read-only): one tensor per (batch size, parameter version).  A captured HIP graph
        reads it by address, so entries are only dropped wholesale and graphs hold the entries they were captured with
        (graphs.cache_refs)."""
        qf = self.query_feat.weight
        qkey = (B, str(dev), qf.data_ptr(), qf._version)
        q0 = getattr(self, "_q0", None)
        if q0 is None or len(q0) > 32:
            q0 = self._q0 = {}
        if qkey not in q0:
            q0[qkey] = qf[None].expand(B, -1, -1).contiguous()
        return q0[qkey]

    @torch.no_grad()
    def forward(self, x, mask_features, mask=None, *, final_topk=0):
        """``final_topk`` = K > 0 (MeanShiftMaskFormer.inference): the final mask step runs on the K (query, class) pairs
        instance_inference keeps (PM:461-497) and the result carries them as "topk"; 0: all queries, like the reference."""
        assert len(x) == self.num_feature_levels
        del mask
        final_topk = int(final_topk)
        B = x[0].shape[0]
        dev = x[0].device
        E = self.query_feat.weight.shape[1]
        src, pos, sizes, xs = [], [], [], []
        for i in range(self.num_feature_levels):
            h, w = x[i].shape[-2:]
            sizes.append((int(h), int(w)))
            # token-major (channels_last) level maps, as the pixel decoder returns them, are consumed as they are
            xs.append(x[i] if ops.is_token_major(x[i]) else x[i].contiguous())
        kv_all = None
        fkv = None
        if self.fold_kv:
            kv_w, kv_c = self._folded_kv(sizes, dev)
            fkv = self._fused_kv_plan(xs, sizes, kv_w, kv_c)
            kv_bytes = (2 if self.attention_dtype == "bf16" else 4) * B * kv_w[0].shape[0] * sum(sizes[i % self.num_feature_levels][0] * sizes[i % self.num_feature_levels][1]
                                                      for i in range(self.num_layers))
            if (self.batched_kv and kv_bytes <= (1 << 30) and all(xl.shape[1] == 64 for xl in xs)
                    and kv_w[0].shape[0] in (256, 512)):       # all layers' K/V live at once: only while that stays small (beyond ~1 GiB a
                # layer's K/V is long out of the caches when its attention reads it: configs[4] at batch 4 is 1 % faster layer by layer)
                # (a launch takes up to 16 jobs: the 20 layers of configs[4] are two launches)
                kv_all = [None] * self.num_layers
                todo = [i for i in range(self.num_layers) if fkv is None or fkv["layers"][i] is None]      # (fused layers project inside their attention)
                for j0 in range(0, len(todo), 16):
                    jobs = todo[j0:j0 + 16]
                    jc = self._uniform_constants(kv_c, jobs)
                    for i, kv in zip(jobs, ops.kv_project_multi([xs[i % self.num_feature_levels] for i in jobs], [kv_w[i] for i in jobs],
                                                   [c for c, _ in jc],
                                                   out_dtype=torch.bfloat16 if self.attention_dtype == "bf16" else torch.float32,
                                                   split=self.kv_split and self.attention_dtype != "bf16",
                                                   cmat_widths=[cw for _, cw in jc],
                                                   keys_f16=self._keys_f16())):
                        kv_all[i] = kv
        else:
            for i in range(self.num_feature_levels):
                pos.append(self._pos_tokens(*sizes[i], dev))
                if isinstance(self.input_proj[i], nn.Conv2d):
                    wt = self.input_proj[i].weight.view(E, -1)
                    bias = self.input_proj[i].bias + self.level_embed.weight[i]          # DEC:575
                    src.append(ops.conv1x1_nchw_to_tokens(xs[i].contiguous(), wt, bias.contiguous()))
                else:
                    src.append(ops.transpose_last2(xs[i].contiguous().flatten(2)) + self.level_embed.weight[i])
        if isinstance(mask_features, ConvFoldedMaskFeatures):
            # the UCN path's factored mask features: taken when every cross attention reads bit-packed masks (fused K/V attention on
            # every layer) and no intermediate logits are asked for; else the literal tensor
            Qn = self.query_feat.weight.shape[0]
            if (self.folded_mask_features and self.fused_tails and self.fold_kv and not self.aux_outputs and self.num_layers > 0
                    and self.mask_step_dtype in ("bf16", "f16") and Qn <= 112 and fkv is not None and all(l is not None for l in fkv["layers"])
                    and self.num_feature_levels == 1 and mask_features.x is x[0] and tuple(mask_features.weight.shape[1:]) == (64, 3, 3)
                    and mask_features.weight.shape[0] == E):
                return self._forward_fused(xs, sizes, kv_w, kv_c, mask_features, self._initial_queries(B, dev), self.query_embed.weight, kv_all,
                                           final_topk, fkv)
            mask_features = mask_features.tensor()
        folded = isinstance(mask_features, FoldedMaskFeatures)
        if folded and not (self.folded_mask_features and self.fused_tails and self.fold_kv and mask_features.act.shape[1] % 32 == 0
                           and mask_features.act.shape[1] < self.query_feat.weight.shape[1]):
            mask_features, folded = mask_features.tensor(), False           # literal order: materialise (B, mask_dim, H, W)
        if not folded:
            mask_features = mask_features.contiguous()
        if self.mask_step_dtype not in ("f32", "bf16", "f16", "f32_split"):
            raise ValueError("mask_step_dtype must be 'f32', 'bf16', 'f16' or 'f32_split'")
        mf_planes = mask_features.act if folded else mask_features
        # the default inference plan never runs the full-resolution mask kernel on all queries (attention masks at key resolution,
        # the final step on the top-K embeddings with the fp32 kernel): no packed copy of the activation is needed then
        lean = (folded and self.fused_tails and self.fold_kv and not self.aux_outputs and self.pooled_attention_masks and self.num_layers > 0
                and mf_planes.shape[1] == 64 and 0 < final_topk < self.query_feat.weight.shape[0]
                and len(self._poolable_sizes(mf_planes, sizes)) == len(set((int(a), int(b)) for a, b in sizes)))
        self._packed_mf = ops.pack_mask_features_bf16(mf_planes, f16=self.mask_step_dtype == "f16") \
            if (self.mask_step_dtype in ("bf16", "f16") and not lean) else None
        # f32_split: the folded 64-channel step as exact three-term bf16 splits on the bf16 matrix pipe (fp32-accurate); the
        # literal 256-channel form keeps the fp32 MFMA kernel
        self._packed_mf_split = ops.pack_mask_features_split(mf_planes) \
            if (self.mask_step_dtype == "f32_split" and folded and mf_planes.shape[1] == 64 and not lean) else None
        qpos = self.query_embed.weight
        out = self._initial_queries(B, dev)
        full = self.aux_outputs
        L = self.num_layers
        pred_cls, pred_mask = [], []
        if self.fused_tails and self.fold_kv:
            return self._forward_fused(xs, sizes, kv_w, kv_c, mask_features, out, qpos, kv_all, final_topk, fkv)
        d = ops.layernorm(out, self.decoder_norm.weight, self.decoder_norm.bias)
        cls, m, attn, row_any = self._heads(d, mask_features, sizes[0], full or L == 0, full or L == 0)
        pred_cls.append(cls)
        pred_mask.append(m)
        for i in range(L):
            lvl = i % self.num_feature_levels                                     # DEC:608
            ca = self.transformer_cross_attention_layers[i]
            if self.fold_kv:
                if kv_all is not None:
                    kv = kv_all[i]
                else:
                    kv = ops.kv_project(xs[lvl], kv_w[i], *kv_c[i])       # (B, hw, 2E) = [K | V]
                t2 = ca.meanshift_attn.attend(out, None, None, query_pos=qpos, masked=attn, row_any=row_any,
                                              kv=(kv[..., :E], kv[..., E:]))
            else:
                t2 = ca.meanshift_attn.attend(out, src[lvl], src[lvl], query_pos=qpos, key_pos=pos[lvl],
                                              masked=attn, row_any=row_any)
            out = ops.layernorm(out, ca.norm.weight, ca.norm.bias, parts=t2[None])
            sa = self.transformer_self_attention_layers[i]
            w, b = sa.self_attn.in_proj_weight, sa.self_attn.in_proj_bias
            qk = ops.gemm(out, w[:2 * E], b[:2 * E], a2=qpos)                      # q and k share tgt + query_pos
            v = ops.gemm(out, w[2 * E:], b[2 * E:])
            o = ops.hypersphere_attention(qk[..., :E], qk[..., E:], v, self.num_heads, kappa=float(KAPPA))
            t2 = ops.gemm(o, sa.self_attn.out_proj.weight, sa.self_attn.out_proj.bias)
            out = ops.layernorm(out, sa.norm.weight, sa.norm.bias, parts=t2[None])
            ff = self.transformer_ffn_layers[i]
            hdn = ops.gemm(out, ff.linear1.weight, ff.linear1.bias, act="relu")
            parts = ops.gemm(hdn, ff.linear2.weight, split_k=8 if ff.linear2.weight.shape[1] >= 1024 else 1)
            if parts.dim() == 3:
                parts = parts[None]
            out, d = ops.layernorm(out, ff.norm.weight, ff.norm.bias, parts=parts, bias=ff.linear2.bias,
                                   l2norm=self.decoder_block_norm, g2=self.decoder_norm.weight,
                                   b2=self.decoder_norm.bias)
            last = i == L - 1
            tgt = None if (last and not full) else sizes[(i + 1) % self.num_feature_levels]
            cls, m, attn, row_any = self._heads(d, mask_features, tgt, full or last, full or last)
            pred_cls.append(cls)
            pred_mask.append(m)
        res = {"pred_logits": pred_cls[-1], "pred_masks": pred_mask[-1], "aux_outputs": []}
        if full:
            res["aux_outputs"] = [{"pred_logits": a, "pred_masks": b} for a, b in zip(pred_cls[:-1], pred_mask[:-1])]
        return res


class PretrainedMeanShiftTransformerDecoder(MeanShiftTransformerDecoder):
    """meanshiftformer_transformer_decoder.py:697-1048: the same decoder over ONE feature level -- every
    pixel of the full-resolution 64-channel UCN embedding is a key, and the attention mask has the
    resolution of the mask logits themselves (target size == mask size, so the bilinear resize is the
    identity).  The kernels stream keys blockwise, so 307 200 keys need nothing new."""
    NUM_FEATURE_LEVELS = 1


class SimpleBasePixelDecoder(PlanAttributes, nn.Module):
    """pixel_decoder/fpn.py:161-290: passes the backbone embedding through and, when mask_dim != 64,
    derives mask_features with one 3x3 convolution (with bias, no norm)."""

    def __init__(self, input_shape, *, conv_dim, mask_dim, norm=None):
        super().__init__()
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.mask_dim = mask_dim
        self.conv_dim = conv_dim
        self.precision = "f32"             # "bf16" (head.set_precision): the mask_features convolution in the low-precision form
        self.lp_operands = "bf16"          # ... with bf16 or IEEE-half ("f16") operands
        self.fold_mask_conv = True         # 16-bit plans: hand the decoder the factored form (ConvFoldedMaskFeatures) when it asks for it
        if mask_dim != 64:
            self.mask_features = nn.Conv2d(conv_dim, mask_dim, kernel_size=3, stride=1, padding=1)
        self.maskformer_num_feature_levels = 1

    @classmethod
    def from_config(cls, cfg, input_shape):
        sh = cfg.MODEL.SEM_SEG_HEAD
        return dict(input_shape={k: v for k, v in input_shape.items() if k in sh.IN_FEATURES},
                    conv_dim=sh.CONVS_DIM, mask_dim=sh.MASK_DIM, norm=sh.NORM)

    @torch.no_grad()
    def forward_features(self, features, folded=False):
        """``folded`` (16-bit plans, a 64-channel feature with W % 16 == 0): mask_features come back in factored form
        (ConvFoldedMaskFeatures) -- the convolution is folded into the decoder's query embedding and never run."""
        multi_scale_features = []
        y = None
        for f in self.in_features[::-1]:
            y = features[f]
            if len(multi_scale_features) < self.maskformer_num_feature_levels:
                multi_scale_features.append(y)
        if self.mask_dim == 64:
            return y, None, multi_scale_features
        B, C, H, W = y.shape
        lp = getattr(self, "precision", "f32") == "bf16"

        def literal():
            tok = ops.transpose_last2(y.contiguous().view(B, C, H * W))                       # NHWC tokens
            w = self.mask_features.weight.permute(0, 2, 3, 1).reshape(self.mask_dim, 9 * C).contiguous()
            mf = ops.conv3x3_tokens_to_nchw(tok, w, self.mask_features.bias, int(H), int(W),
                                            bf16=("f16" if getattr(self, "lp_operands", "bf16") == "f16" else True) if lp else False)
            return mf.view(B, self.mask_dim, H, W)

        if folded and lp and getattr(self, "fold_mask_conv", True) and y.is_cuda and C == 64 and W % 16 == 0 and H * W * 128 < (1 << 32) - 256:
            return ConvFoldedMaskFeatures(y, self.mask_features.weight, self.mask_features.bias, literal), None, multi_scale_features
        return literal(), None, multi_scale_features


# ----------------------------------------------------------------------------------------------
class MSDeformAttn(nn.Module):
    """ops/modules/ms_deform_attn.py:34-125 with the same parameters; forward takes the reference's
    arguments.  The encoder calls ``forward_encoder`` (fused sampling arithmetic) instead."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.constant_(self.sampling_offsets.weight.data, 0.)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(
            1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.)
        nn.init.constant_(self.attention_weights.bias.data, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.)

    def _proj_weights(self):
        return (torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0).contiguous(),
                torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0).contiguous())

    def forward_encoder(self, src, lvl_pos, spatial_shapes, level_start_index):
        """src (N,S,C); query = src + lvl_pos; reference points = pixel centres."""
        value = ops.gemm(src, self.value_proj.weight, self.value_proj.bias)
        w, b = self._proj_weights()
        proj = ops.gemm(src, w, b, a2=lvl_pos)
        out = ops.ms_deform_attn_encoder(value, spatial_shapes, level_start_index, proj, self.n_heads, self.n_points)
        return ops.gemm(out, self.output_proj.weight, self.output_proj.bias)

    @torch.no_grad()
    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        if input_padding_mask is not None:
            raise NotImplementedError("padding masks are not used by the MSMFormer configs")
        if reference_points.shape[-1] != 2:
            raise NotImplementedError("only 2-d reference points")
        N, Lq, _ = query.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        value = ops.gemm(input_flatten.contiguous(), self.value_proj.weight, self.value_proj.bias)
        value = value.view(N, -1, M, self.d_model // M)
        q = query.contiguous()
        off = ops.gemm(q, self.sampling_offsets.weight, self.sampling_offsets.bias).view(N, Lq, M, L, P, 2)
        aw = ops.gemm(q, self.attention_weights.weight, self.attention_weights.bias).view(N, Lq, M, L * P)
        # softmax over the L*P logits and loc = ref + off / (W_l, H_l) (ms_deform_attn.py:101-109) in one HIP launch
        loc, aw = ops.msda_locations(off.contiguous(), aw.contiguous(), reference_points.float().contiguous(),
                                     input_spatial_shapes.to(torch.int64).contiguous())
        out = ops.ms_deform_attn(value, input_spatial_shapes, input_level_start_index, loc, aw)
        return ops.gemm(out, self.output_proj.weight, self.output_proj.bias)


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward_tokens(self, src, lvl_pos, spatial_shapes, level_start_index):
        a = self.self_attn.forward_encoder(src, lvl_pos, spatial_shapes, level_start_index)
        src = ops.layernorm(src, self.norm1.weight, self.norm1.bias, parts=a[None])          # MSD:124-126
        h = ops.gemm(src, self.linear1.weight, self.linear1.bias, act="relu")
        f = ops.gemm(h, self.linear2.weight, self.linear2.bias)
        return ops.layernorm(src, self.norm2.weight, self.norm2.bias, parts=f[None])         # MSD:116-118


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, layer_factory, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(layer_factory() for _ in range(num_layers))
        self.num_layers = num_layers


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = MSDeformAttnTransformerEncoder(
            lambda: MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                        num_feature_levels, nhead, enc_n_points),
            num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        nn.init.normal_(self.level_embed)


class _ConvNorm(nn.Conv2d):
    """Parameter container with detectron2's Conv2d naming: .weight (+ .norm.{weight,bias})."""

    def __init__(self, cin, cout, k, bias, norm_channels=None):
        super().__init__(cin, cout, kernel_size=k, padding=k // 2, bias=bias)
        self.norm = nn.GroupNorm(32, norm_channels) if norm_channels else None


class MSDeformAttnPixelDecoder(PlanAttributes, nn.Module):
    """pixel_decoder/msdeformattn.py:164-358 for norm == "GN"."""

    def __init__(self, input_shape, *, transformer_dropout, transformer_nheads, transformer_dim_feedforward,
                 transformer_enc_layers, conv_dim, mask_dim, norm=None, transformer_in_features, common_stride):
        super().__init__()
        if norm != "GN":
            raise NotImplementedError('only SEM_SEG_HEAD.NORM == "GN"')
        tis = {k: v for k, v in input_shape.items() if k in transformer_in_features}
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.feature_strides = [v.stride for k, v in input_shape]
        self.feature_channels = [v.channels for k, v in input_shape]
        tis = sorted(tis.items(), key=lambda x: x[1].stride)
        self.transformer_in_features = [k for k, v in tis]
        t_channels = [v.channels for k, v in tis]
        self.transformer_feature_strides = [v.stride for k, v in tis]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        self.input_proj = nn.ModuleList(
            nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in t_channels[::-1])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.conv_dim = conv_dim
        self.mask_features = nn.Conv2d(conv_dim, mask_dim, kernel_size=1)
        self.maskformer_num_feature_levels = 3
        self.common_stride = common_stride
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(math.log2(stride) - math.log2(self.common_stride))
        if self.num_fpn_levels != 1:
            raise NotImplementedError("exactly one extra FPN level (res2) is supported")
        for idx, cin in enumerate(self.feature_channels[:self.num_fpn_levels]):
            self.add_module("adapter_{}".format(idx + 1), _ConvNorm(cin, conv_dim, 1, False, conv_dim))
            self.add_module("layer_{}".format(idx + 1), _ConvNorm(conv_dim, conv_dim, 3, False, conv_dim))
        self._cache = {}
        self._packed = None
        # "bf16" (BASELINE configs 3 / 5; the reference's low-precision mode is autocast over the whole model,
        # tabletop_train_net_pretrained.py:232): the encoder's token-wise GEMMs run with bf16 MFMA operands and fp32
        # accumulation (msm_encoder_block_lp_fwd); residual stream, LayerNorms, sampling arithmetic and outputs stay fp32.
        # "f32_split": fp32 results on the bf16 matrix pipe -- every operand split exactly into three bf16 terms, six MFMAs
        # per product (msm_encoder_block_split_fwd); as accurate as "f32" (tests measure both against float64), ~30 % faster
        self.precision = "f32"
        self.fused_encoder = True      # False: one GEMM / LayerNorm launch per op (same results up to rounding)
        self.fused_front = True        # False: input projections and layer 0's projections as separate GEMM / GroupNorm launches
        # True: a layer's gather computes its own [sampling_offsets | attention_weights] projection on the matrix pipe
        # (ops.ms_deform_attn_encoder_fused) instead of reading the `proj` tensor the previous token kernel wrote (58 MB per
        # layer at B = 8); bitwise the same values.  fp32 plan only (the low-precision / split token kernels keep writing proj).
        # Measured on MI355X at B = 8 (round 3): the token kernel drops from 152 to 123 us per layer, the gather rises from
        # 32 to 59 us (48 MFMAs per wave in front of a latency-bound gather do not overlap with it): neutral in time, 116 MB
        # less HBM traffic per layer -- off by default, DESIGN.md section 4
        self.fused_msda = False
        # bf16 plan only: head-major fp16 value / attention / sampling-projection tensors between the encoder kernels
        # (csrc/enc_lp.hip: msm_encoder_block_hm_fwd + msm_msdeform_attn_enc_lp_fwd; 66 us per layer at B = 8 against 85 with the fp32
        # tensors of round 3).  False: the round-3 kernels (msm_encoder_block_lp_fwd + the fp32 gather)
        self.hm_activations = True
        # operand format of the low-precision plan's FFN stages (head.set_precision("bf16" / "f16")): "f16" = IEEE-half W1 / W2 /
        # activations on v_mfma_f32_16x16x32_f16 (8x smaller roundings at the same rate; csrc/enc_lp.hip, template F16)
        self.lp_operands = "bf16"
        self.lp_input_proj = "deep"         # bf16 plan: the FPN lateral on the bf16 matrix pipe (hi + lo operands, fp32 results; 66 -> 50 us); "deep": the three deep levels too (weight through LDS)
        self.lp_conv3x3 = True              # bf16 plan: the FPN output convolution with bf16 operands (csrc/conv3x3.hip); False: the fp32 kernel
        self.fpn_half_map = True            # f16 operands: the FPN level's GroupNorm output travels to that convolution as IEEE halves (same result bits)
        self.lp_prologue = True             # bf16 plan: the prologue's projections on the bf16 matrix pipe (enc_prologue_hm_kernel)

    def _w3(self):
        """layer_1's 3x3 weight in the implicit-GEMM order (Cout, 3*3*Cin), cached per parameter version."""
        p = self.layer_1.weight
        key = (p.data_ptr(), p._version)
        if getattr(self, "_w3_cache", None) is None or self._w3_cache[0] != key:
            self._w3_cache = (key, p.permute(0, 2, 3, 1).reshape(p.shape[0], -1).contiguous())
        return self._w3_cache[1]

    def _w_lateral(self, lp=False):
        """adapter_1's 1x1 weight in the fragment order of msm_conv1x1_in_f32 (``lp``: the hi + lo bf16 order of msm_conv1x1_in_lp),
        cached per parameter version."""
        p = self.adapter_1.weight
        key = (p.data_ptr(), p._version, bool(lp))
        if getattr(self, "_wl_cache", None) is None or self._wl_cache[0] != key:
            pack = ops.pack_conv_in_weight_lp if lp else ops.pack_conv_in_weight
            self._wl_cache = (key, pack(p.view(p.shape[0], -1)))
        return self._wl_cache[1]

    def _lp_input_proj(self, channels):
        """The bf16 plan runs the FPN lateral on the bf16 matrix pipe (hi + lo operands, fp32 results): 64 output channels,
        input channels a multiple of 256."""
        return self.precision == "bf16" and self.lp_input_proj and self.conv_dim == 64 and all(int(c) % 256 == 0 for c in channels)

    def _use_fused_msda(self, device):
        """The gather computes its own sampling projection: fp32 plan, the shipped geometry (64 channels, 8 heads, 3 levels x 4
        points), 16-byte-aligned token buffers."""
        if not (self.fused_msda and self.fused_encoder and self.precision == "f32" and self.conv_dim == 64):
            return False
        self._packed_encoder(device)
        return self._packed[3] is not None and len(self.transformer_in_features) == 3

    def _use_hm(self):
        """The bf16 plan's head-major bf16 activations: the shipped geometry (64 channels, 8 heads, 3 levels x 4 points)."""
        layers = self.transformer.encoder.layers
        return (self.precision == "bf16" and self.hm_activations and self.fused_encoder and self.conv_dim == 64
                and len(self.transformer_in_features) == 3
                and all(ly.self_attn.d_model == 64 and ly.self_attn.n_heads == 8 and ly.self_attn.n_levels == 3 and ly.self_attn.n_points == 4
                        and ly.linear1.out_features % 32 == 0 for ly in layers))

    def _packed_encoder(self, device):
        """Weight streams of the fused encoder kernel, rebuilt only when a parameter changes."""
        layers = self.transformer.encoder.layers
        if self.precision not in ("f32", "f32_split", "bf16"):
            raise ValueError("precision must be 'f32', 'f32_split' or 'bf16'")
        if getattr(self, "_enc_params", None) is None:
            self._enc_params = TensorList.of(self, "transformer.encoder")
        hm = self._use_hm()
        key = (str(device), self.precision, hm, self.lp_operands) + version_key(self._enc_params())
        if self._packed is None or self._packed[0] != key:
            out = []
            for l, layer in enumerate(layers):
                nxt = layers[l + 1].self_attn if l + 1 < len(layers) else None
                a = layer.self_attn
                if hm:
                    wv = wp = bv = bp = None
                    if nxt is not None:
                        wv, bv = nxt.value_proj.weight, nxt.value_proj.bias
                        wp, bp = nxt._proj_weights()
                    stream = ops.pack_encoder_block_hm(a.output_proj.weight, layer.linear1.weight, layer.linear2.weight, wv, wp,
                                                       ffn_f16=self.lp_operands == "f16")
                    small = ops.pack_encoder_block_hm_small(a.output_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.bias,
                                                            layer.linear2.bias, layer.norm2.weight, layer.norm2.bias, bv, bp)
                    out.append((stream, small, layer.linear1.out_features, 0))
                    continue
                wv = wp = None
                smalls = [a.output_proj.bias, layer.norm1.weight, layer.norm1.bias, layer.linear1.bias, layer.linear2.bias,
                          layer.norm2.weight, layer.norm2.bias]
                if nxt is not None:
                    wv = nxt.value_proj.weight
                    wp, bp = nxt._proj_weights()
                    smalls += [nxt.value_proj.bias, bp]
                else:
                    smalls += [torch.zeros(64, device=device), torch.zeros(a.sampling_offsets.out_features + a.attention_weights.out_features, device=device)]
                pack = {"f32": ops.pack_encoder_block, "f32_split": ops.pack_encoder_block_split, "bf16": ops.pack_encoder_block_lp}[self.precision]
                stream = pack(a.output_proj.weight, layer.linear1.weight, layer.linear2.weight, wv, wp)
                pw = a.sampling_offsets.out_features + a.attention_weights.out_features
                out.append((stream, torch.cat([t.reshape(-1) for t in smalls]).contiguous(), layer.linear1.out_features, pw))
            msda = None
            if all(ly.self_attn.d_model == 64 and ly.self_attn.n_heads == 8 and ly.self_attn.n_levels * ly.self_attn.n_points == 12
                   and ly.self_attn.n_points == 4 for ly in layers):
                msda = [ops.pack_msda_proj(*ly.self_attn._proj_weights(), ly.self_attn.n_heads, ly.self_attn.n_levels, ly.self_attn.n_points)
                        for ly in layers]
            self._packed = (key, out, layers[0].self_attn._proj_weights(), msda)
        return self._packed[1]

    @classmethod
    def from_config(cls, cfg, input_shape):
        sh, mf = cfg.MODEL.SEM_SEG_HEAD, cfg.MODEL.MASK_FORMER
        return dict(input_shape={k: v for k, v in input_shape.items() if k in sh.IN_FEATURES},
                    conv_dim=sh.CONVS_DIM, mask_dim=sh.MASK_DIM, norm=sh.NORM, transformer_dropout=mf.DROPOUT,
                    transformer_nheads=mf.NHEADS, transformer_dim_feedforward=1024,
                    transformer_enc_layers=sh.TRANSFORMER_ENC_LAYERS,
                    transformer_in_features=sh.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES,
                    common_stride=sh.COMMON_STRIDE)

    def _geometry(self, shapes, device):
        key = (tuple(shapes), str(device))
        if key not in self._cache:
            ss = torch.tensor(shapes, dtype=torch.int64, device=device)
            starts = torch.tensor([0] + list(torch.tensor([h * w for h, w in shapes]).cumsum(0)[:-1].tolist()),
                                  dtype=torch.int64, device=device)
            pos = [ops.pos_embed_sine(h, w, self.pe_layer.num_pos_feats, device, layout="tokens",
                                      add_c=self.transformer.level_embed[l].contiguous())           # MSD:75
                   for l, (h, w) in enumerate(shapes)]
            self._cache[key] = (ss, starts, torch.cat(pos, 0).contiguous())
        return self._cache[key]

    def _packed_front(self, device):
        """Fragment-order input_proj weights, GroupNorm parameters and the prologue weight stream (layer 0's value /
        sampling projections), rebuilt only when one of those parameters changes."""
        a0 = self.transformer.encoder.layers[0].self_attn
        params = [p for m in self.input_proj for p in m.parameters()] + list(a0.value_proj.parameters()) + \
            list(a0.sampling_offsets.parameters()) + list(a0.attention_weights.parameters())
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in params)
        if getattr(self, "_front", None) is None or self._front[0] != key:
            C = self.conv_dim
            wpk = [ops.pack_conv_in_weight(m[0].weight.view(C, -1)) for m in self.input_proj]
            gnp = torch.stack([torch.stack([m[1].weight, m[1].bias]) for m in self.input_proj]).contiguous()
            wp, bp = a0._proj_weights()
            stream = ops.pack_encoder_prologue(a0.value_proj.weight, wp)
            small = torch.cat([a0.value_proj.bias, bp]).contiguous()
            # the bf16 plan's prologue (hi + lo bf16 fragment blocks): packed when that plan's geometry holds
            hm = ops.pack_encoder_prologue_hm(a0.value_proj.weight, wp, a0.value_proj.bias, bp) \
                if (C == 64 and tuple(wp.shape) == (288, 64) and a0.n_heads == 8) else None
            # the 16-bit plans' deep levels: hi + lo bf16 fragment order (ops.conv1x1_in_multi(lp="wide")), packed when the shapes allow
            wlp = [ops.pack_conv_in_weight_lp(m[0].weight.view(C, -1)) for m in self.input_proj] \
                if C == 64 and all(m[0].weight.shape[1] % 256 == 0 for m in self.input_proj) else None
            self._front = (key, wpk, gnp, stream, small, wp.shape[0], hm, wlp)
        return self._front[1:]

    def _encode(self, features):
        """Input projections + the six encoder layers.  Returns the token buffer (B, S, C) with the levels concatenated
        coarse to fine, their (h, w) shapes and the pre-zeroed moment buffers of the two FPN GroupNorms (or Nones)."""
        C = self.conv_dim
        levels = [features[f].float().contiguous() for f in self.transformer_in_features[::-1]]      # res5, res4, res3
        B = levels[0].shape[0]
        shapes = [(int(x.shape[2]), int(x.shape[3])) for x in levels]
        dev = levels[0].device
        ss, starts, lvl_pos = self._geometry(shapes, dev)
        S_tok = sum(h * w for h, w in shapes)
        layers = self.transformer.encoder.layers
        gns = [m[1] for m in self.input_proj]
        front = (self.fused_encoder and self.fused_front and C == 64 and len(levels) <= 4 and S_tok >= 86
                 and all(x.shape[1] % 128 == 0 and (x.shape[2] * x.shape[3]) % 4 == 0 for x in levels)
                 and all(g.num_groups == gns[0].num_groups and g.eps == gns[0].eps for g in gns))
        value = proj = None
        fpn_stats = (None, None)
        if front:
            # input projections straight into the concatenated token buffer with their GroupNorm moments as a
            # by-product, then ONE prologue pass: GroupNorm, layer 0's value projection and sampling projections
            wpk, gnp, pstream, psmall, pw, phm, wlp = self._packed_front(dev)
            src = torch.empty((B, S_tok, C), device=dev, dtype=torch.float32)
            stats = torch.zeros((len(levels) + 2, B, C, 2), device=dev, dtype=torch.float64)      # + the two FPN GroupNorms
            fpn_stats = (stats[len(levels)], stats[len(levels) + 1])
            # (fp32 MFMA kernel in every plan: on the bf16 pipe these three deep-K levels are bound by their weight traffic at the same
            # 66 us -- DESIGN.md section 4a, k31; the bf16 plan moves the shallow lateral, whose weight fits LDS)
            # round 6: with the packed weight broadcast through LDS (eight-wave workgroups over adjacent tiles x K slices) the bf16 pipe wins
            # where that shape gives every CU a workgroup -- lp_input_proj = "deep"; smaller batches keep the fp32 kernel
            if self.precision == "bf16" and self.lp_input_proj == "deep" and wlp is not None and B * S_tok >= 16384:
                ops.conv1x1_in_multi(levels, wlp, [m[0].bias for m in self.input_proj], src, stats[:len(levels)], stats_cleared=True, lp="wide")
            else:
                ops.conv1x1_in_multi(levels, wpk, [m[0].bias for m in self.input_proj], src, stats[:len(levels)], stats_cleared=True)
            a0 = layers[0].self_attn
            bounds = [0]
            for h, w in shapes:
                bounds.append(bounds[-1] + h * w)
            fuse0 = self._use_fused_msda(dev)
            if self._use_hm() and phm is not None and self.lp_prologue:
                # the bf16 plan: the two projections on the bf16 matrix pipe (csrc/enc_lp.hip, enc_prologue_hm_kernel)
                src, value, proj = ops.encoder_prologue_hm(src, stats[:len(levels)], gnp, bounds, phm[0], phm[1], lvl_pos,
                                                           groups=gns[0].num_groups, eps=gns[0].eps)
            else:
                src, value, proj = ops.encoder_prologue(src, stats[:len(levels)], gnp, bounds, pstream, psmall[:64] if fuse0 else psmall, lvl_pos,
                                                        0 if fuse0 else pw, groups=gns[0].num_groups, eps=gns[0].eps, value_heads=a0.n_heads,
                                                        bf16_hm=self._use_hm())
        else:
            toks = []
            for idx, x in enumerate(levels):
                conv, gn = self.input_proj[idx][0], self.input_proj[idx][1]
                t = ops.conv1x1_nchw_to_tokens(x, conv.weight.view(C, -1), conv.bias)
                toks.append(ops.groupnorm_tokens(t, gn.weight, gn.bias, shapes[idx][0], shapes[idx][1], groups=gn.num_groups, eps=gn.eps))
            src = torch.cat(toks, 1).contiguous()                                     # (B,S,C)
        if self.fused_encoder and C == 64:
            # layer l = MSDeformAttn gather + ONE fused token-wise kernel that also emits layer l+1's
            # value / sampling projections (the 1024-wide FFN activation never leaves registers)
            packed = self._packed_encoder(dev)
            fuse = self._use_fused_msda(dev)
            if value is None:
                a0 = layers[0].self_attn
                value = ops.value_to_head_major(ops.gemm(src, a0.value_proj.weight, a0.value_proj.bias), a0.n_heads)
                if not fuse:
                    w, b = self._packed[2]
                    proj = ops.gemm(src, w, b, a2=lvl_pos)
            if self._use_hm():
                # bf16 plan: value / attention / sampling projection travel between the kernels as head-major fp16
                # (layer 0's come from the fp32 prologue: one conversion each)
                if value.dtype != torch.float16:                     # (the unfused front end: fp32 GEMM results, converted once)
                    value = ops.to_f16(value if value.dim() == 4 else ops.value_to_head_major(value, 8))
                    proj = ops.proj_to_head_major_records(proj)
                for l, layer in enumerate(layers):
                    attn = ops.ms_deform_attn_encoder_lp(value, ss, starts, proj, layer.self_attn.n_points)
                    stream, small, d_ffn, _ = packed[l]
                    src, value, proj = ops.encoder_block_hm(attn, src, stream, small, d_ffn, pos=lvl_pos, want_next=l + 1 < len(layers),
                                                            eps=layer.norm1.eps, ffn_f16=self.lp_operands == "f16")
                return src, shapes, fpn_stats
            for l, layer in enumerate(layers):
                if fuse:
                    attn = ops.ms_deform_attn_encoder_fused(value, ss, starts, src, lvl_pos, *self._packed[3][l], layer.self_attn.n_points)
                else:
                    attn = ops.ms_deform_attn_encoder(value, ss, starts, proj, layer.self_attn.n_heads, layer.self_attn.n_points)
                stream, small, d_ffn, pw = packed[l]
                if fuse:
                    pw = 0                         # the block emits the next layer's value only
                # layers 1.. read a head-major value (written so by the previous block): 64-byte instead of 32-byte taps
                block = {"f32": ops.encoder_block, "f32_split": ops.encoder_block_split, "bf16": ops.encoder_block_lp}[self.precision]
                src, value, proj = block(attn, src, stream, small, d_ffn, pw, pos=lvl_pos, tokens_per_image=S_tok,
                                                     want_next=l + 1 < len(layers), eps=layer.norm1.eps,
                                                     value_heads=layers[l + 1].self_attn.n_heads if l + 1 < len(layers) else 0)
        else:
            for layer in layers:
                src = layer.forward_tokens(src, lvl_pos, ss, starts)
        return src, shapes, fpn_stats

    def _fpn_mask_features(self, features, up_tok, up_hw, fpn_stats, folded=False):
        """The one FPN level on res2 and the mask_features convolution (MSD:343-358); up_tok: the finest encoder level as a
        token-range view of the encoder's buffer."""
        C = self.conv_dim
        B = up_tok.shape[0]
        # one FPN level on the highest-resolution backbone feature (MSD:343-351)
        x = features[self.in_features[0]].float().contiguous()
        H, W = int(x.shape[2]), int(x.shape[3])
        split3 = C == 64 and self.precision == "f32_split"       # the 3x3 convolution on the bf16 matrix pipe
        # "f16" plan: the GroupNorm writes the halves the convolution would round to, the convolution keeps a unit's loads in flight at once
        half_map = C == 64 and self.precision == "bf16" and self.lp_conv3x3 and self.lp_operands == "f16" and self.fpn_half_map
        if C == 64 and x.shape[1] % 128 == 0 and x.shape[1] <= 384 and (H * W) % 4 == 0 and B * H * W >= 32 * 1024:
            # shallow-K input-projection kernel: the GroupNorm moments come out of its epilogue (no moments pass over lat)
            lp = self._lp_input_proj([x.shape[1]])
            lat, lat_stats = ops.conv1x1_in(x, self._w_lateral(lp), None, stats=fpn_stats[0], stats_cleared=fpn_stats[0] is not None, lp=lp)
            y = ops.groupnorm_tokens(lat, self.adapter_1.norm.weight, self.adapter_1.norm.bias, H, W, groups=32,
                                     up=up_tok, up_hw=up_hw, eps=self.adapter_1.norm.eps, stats=lat_stats, stats_ready=True,
                                     split_planes=split3, out_f16=half_map)
        else:
            lat = ops.conv1x1_nchw_to_tokens(x, self.adapter_1.weight.view(C, -1), None)
            y = ops.groupnorm_tokens(lat, self.adapter_1.norm.weight, self.adapter_1.norm.bias, H, W, groups=32,
                                     up=up_tok, up_hw=up_hw, eps=self.adapter_1.norm.eps, stats=fpn_stats[0], split_planes=split3,
                                     out_f16=half_map)
        y_stats = None
        if C == 64:
            # weight-stationary 3x3 kernel; the moments of layer_1's GroupNorm come out of its epilogue.  f32_split: the
            # GroupNorm above wrote its result as three bf16 planes, the convolution multiplies exact three-term splits
            y, y_stats = ops.conv3x3_c64(y, self._w3(), H, W, stats=fpn_stats[1], stats_cleared=fpn_stats[1] is not None,
                                         bf16=(self.lp_operands == "f16" and "f16" or True) if (self.precision == "bf16" and self.lp_conv3x3) else False,
                                         split=split3)
        else:
            y = ops.conv3x3_tokens(y, self._w3(), H, W)
        wm = self.mask_features.weight.view(self.mask_dim, C)
        if C == 64 and self.mask_dim in (256, 512) and (H * W) % 4 == 0 and (B <= 64 or folded):
            # layer_1's GroupNorm + ReLU is applied to the operand fragments of the mask_features convolution
            gn = (y_stats, self.layer_1.norm.weight, self.layer_1.norm.bias, 32, self.layer_1.norm.eps)
            if B <= 64:
                literal = lambda: ops.tokens_proj_nchw(y, wm, self.mask_features.bias, gn=gn, relu=True).view(B, self.mask_dim, H, W)
            else:       # beyond the fused kernel's per-image GroupNorm table (the second stage of the two-stage harness: ~170 crops)
                literal = lambda: ops.conv1x1_tokens_to_nchw(
                    ops.groupnorm_tokens(y, self.layer_1.norm.weight, self.layer_1.norm.bias, H, W, groups=32, relu=True,
                                         eps=self.layer_1.norm.eps), wm, self.mask_features.bias).view(B, self.mask_dim, H, W)
            if folded:
                # hand over the factored form: the 64-channel activation as NCHW planes + the 1x1 weight (FoldedMaskFeatures)
                act = ops.groupnorm_nchw(y, y_stats, self.layer_1.norm.weight, self.layer_1.norm.bias, groups=32,
                                         eps=self.layer_1.norm.eps, relu=True).view(B, C, H, W)
                return FoldedMaskFeatures(act, wm, self.mask_features.bias, literal)
            mask_features = literal()
        else:
            y = ops.groupnorm_tokens(y, self.layer_1.norm.weight, self.layer_1.norm.bias, H, W, groups=32, relu=True,
                                     eps=self.layer_1.norm.eps)
            mask_features = ops.conv1x1_tokens_to_nchw(y, wm, self.mask_features.bias).view(B, self.mask_dim, H, W)
        return mask_features

    @torch.no_grad()
    def forward_features(self, features, folded=False):
        """Returns (mask_features, encoder level 0, multi-scale features) like the reference.  ``folded=True`` (asked for by a
        head whose predictor understands it) returns mask_features as FoldedMaskFeatures instead of a (B, mask_dim, H, W)
        tensor."""
        C = self.conv_dim
        src, shapes, fpn_stats = self._encode(features)
        B = src.shape[0]
        # multi-scale outputs: NCHW-shaped VIEWS of the token buffer (torch channels_last strides) -- no slice copies,
        # no transposes; the decoder's K/V projection reads this layout directly
        out, o = [], 0
        for (h, w) in shapes:
            out.append(src[:, o:o + h * w].view(B, h, w, C).permute(0, 3, 1, 2))
            o += h * w
        up_tok = src[:, o - shapes[-1][0] * shapes[-1][1]:]                       # finest level, source of the FPN upsample (a view)
        mask_features = self._fpn_mask_features(features, up_tok, shapes[-1], fpn_stats, folded)
        return mask_features, out[0], out[:self.maskformer_num_feature_levels]
