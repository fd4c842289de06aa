"""Thin torch-tensor front end over the C ABI (include/msm_hip.h).

torch is used for device memory and streams only: every function checks its arguments, allocates
the outputs with torch.empty on the inputs' device and launches the HIP kernels of libmsm_hip.so
on torch's current stream.  Tensors must be fp32 and live on a ROCm device; there is no CPU path.
"""
import ctypes

import torch

from ._lib import check, lib

KAPPA = 30.0  # attention_util.py:26


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's current HIP stream of the current device as a void*.  torch.cuda.current_stream() costs ~8 us of Python
    per call (device-index resolution + a Stream object) -- a tenth of a small-batch forward; the raw accessor the
    public call itself ends in costs ~0.3 us."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be on the GPU (no CPU path in unseenobjectswithmeanshift_amd)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")


def _c(t, name, dtype=torch.float32):
    _chk(t, name, dtype)
    if t is not None and not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


# ----------------------------------------------------------------------------------------------
def gemm(a, w, bias=None, *, a2=None, act=None, out=None, split_k=1):
    """out[..., n] = act((a + a2) @ w.T + bias) for row-major a (..., K) and w (N, K).
    split_k > 1 returns raw partial sums of shape (split_k, ..., N) (bias/act must be None)."""
    _c(a, "a"), _c(w, "w"), _c(bias, "bias"), _c(a2, "a2")
    K = a.shape[-1]
    N = w.shape[0]
    M = a.numel() // K
    a2_sb = 0
    if a2 is not None:
        if a2.shape == a.shape:
            batch, Mb = 1, M
        else:
            # a2 broadcast over the leading batch dim of a: a (B, L, K), a2 (L, K)
            if a.dim() != 3 or tuple(a2.shape) != tuple(a.shape[1:]):
                raise RuntimeError("a2 must match a or a[0]")
            batch, Mb = a.shape[0], a.shape[1]
    else:
        batch, Mb = 1, M
    lead = a.shape[:-1]
    if split_k > 1:
        if bias is not None or act is not None:
            raise RuntimeError("split_k output is raw: bias/act are applied by the consumer")
        out = torch.empty((split_k,) + tuple(lead) + (N,), device=a.device, dtype=torch.float32)
    elif out is None:
        out = torch.empty(tuple(lead) + (N,), device=a.device, dtype=torch.float32)
    rc = lib().msm_gemm_f32(_p(a), _p(a2), _p(w), _p(bias), _p(out), Mb, N, K, batch,
                            K, 1, Mb * K, a2_sb, 0, N, 1, Mb * N, M * N,
                            0, 0, 0, 0, 1 if bias is not None else 0, 1 if act == "relu" else 0,
                            split_k, _stream())
    check(rc, "msm_gemm_f32")
    return out


def conv1x1_nchw_to_tokens(x, w, bias=None):
    """x (B, Cin, H, W) NCHW -> tokens (B, H*W, Cout) = x^T w^T + bias (a 1x1 Conv2d read through
    the GEMM's M-contiguous A path; no transpose pass).  bias: (Cout,) per channel, or (H*W, Cout) a
    per-position matrix shared by the batch."""
    _c(x, "x"), _c(w, "w"), _c(bias, "bias")
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    HW = H * W
    out = torch.empty((B, HW, Cout), device=x.device, dtype=torch.float32)
    mode = 0 if bias is None else (1 if bias.dim() == 1 else 3)
    if mode == 3 and tuple(bias.shape) != (HW, Cout):
        raise RuntimeError("matrix bias must be (H*W, Cout)")
    rc = lib().msm_gemm_f32(_p(x), None, _p(w), _p(bias), _p(out), HW, Cout, Cin, B,
                            1, HW, Cin * HW, 0, 0, Cout, 1, HW * Cout, 0,
                            0, 0, 0, 0, mode, 0, 1, _stream())
    check(rc, "msm_gemm_f32(conv1x1 nchw)")
    return out


def pack_conv_in_weight(w):
    """(64, Cin) 1x1-convolution weight -> the fragment order msm_conv1x1_in_f32 reads (include/msm_hip.h):
    packed[(((k//8)*4 + o//16)*64 + ((k%8)//2)*16 + o%16)*2 + k%2] = w[o][k]."""
    O, Cin = w.shape
    if O != 64 or Cin % 8:
        raise RuntimeError("pack_conv_in_weight needs a (64, Cin) weight with Cin a multiple of 8")
    return w.reshape(4, 16, Cin // 8, 4, 2).permute(2, 0, 3, 1, 4).contiguous().reshape(-1)


def pack_conv_in_weight_lp(w):
    """(64, Cin) weight -> the hi + lo bf16 fragment order msm_conv1x1_in_lp reads (include/msm_hip.h):
    packed[g][o//16][plane][(k%32)//8][o%16][k%8] = plane(w)[o][k], g = k//32, plane 0 = bf16(w), plane 1 = bf16(w - plane 0)."""
    O, Cin = w.shape
    if O != 64 or Cin % 256:
        raise RuntimeError("pack_conv_in_weight_lp needs a (64, Cin) weight with Cin a multiple of 256")
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    planes = torch.stack([hi, lo])                                                   # (2, 64, Cin)
    return planes.reshape(2, 4, 16, Cin // 32, 4, 8).permute(3, 1, 0, 4, 2, 5).contiguous().reshape(-1)


def conv1x1_in(x, w_packed, bias=None, *, out=None, stats=None, stats_cleared=False, lp=False):
    """Input projection of the pixel decoder: x (B, Cin, H, W) NCHW, w_packed = pack_conv_in_weight(w (64, Cin)) ->
    tokens (B, H*W, 64) = x^T w^T + bias  (``lp``: w_packed = pack_conv_in_weight_lp(w), hi + lo bf16 operands on the bf16
    matrix pipe, fp32 results; Cin a multiple of 256),
    plus the GroupNorm moments of the result.  ``out``: a (B, H*W, 64) view with unit channel stride, row stride 64 and
    any batch stride (e.g. ``buf[:, s:s + H*W]`` of the concatenated token buffer of the encoder); ``stats``: a
    (B, 64, 2) float64 tensor that receives (sum, sum of squares) per (image, channel) -- accumulated into when
    ``stats_cleared`` (the caller zeroed it), else zeroed first.  Returns (out, stats).  Cin must be a multiple of 128
    (conv1x1_nchw_to_tokens + groupnorm_stats cover other shapes)."""
    _c(x, "x"), _c(w_packed, "w_packed", torch.bfloat16 if lp else torch.float32), _c(bias, "bias"), _c(stats, "stats", torch.float64)
    B, Cin, H, W = x.shape
    HW = H * W
    if w_packed.numel() != (128 if lp else 64) * Cin or Cin % (256 if lp else 128) or HW % 4:
        raise RuntimeError("conv1x1_in needs a packed (64, Cin) weight, Cin a multiple of 128 (lp: 256) and H*W a multiple of 4")
    if out is None:
        out = torch.empty((B, HW, 64), device=x.device, dtype=torch.float32)
    _chk(out, "out")
    if tuple(out.shape) != (B, HW, 64) or out.stride(2) != 1 or out.stride(1) != 64 or (B > 1 and out.stride(0) < HW * 64):
        raise RuntimeError("out must be (B, H*W, 64) with strides (>= H*W*64, 64, 1)")
    if stats is None:
        stats = torch.empty((B, 64, 2), device=x.device, dtype=torch.float64)
        stats_cleared = False
    elif tuple(stats.shape) != (B, 64, 2):
        raise RuntimeError("stats must be (B, 64, 2) float64")
    fn = lib().msm_conv1x1_in_lp if lp else lib().msm_conv1x1_in_f32
    rc = fn(_p(x), _p(w_packed), _p(bias), _p(out), out.stride(0) if B > 1 else HW * 64, _p(stats), 1 if stats_cleared else 0, B, Cin, HW,
            _stream())
    check(rc, "msm_conv1x1_in_lp" if lp else "msm_conv1x1_in_f32")
    return out, stats


def conv1x1_in_multi(xs, ws_packed, biases, out, stats, stats_cleared=False, lp=False):
    """conv1x1_in for up to four levels in one launch (``lp``: ws_packed from pack_conv_in_weight_lp, the bf16 matrix pipe; ``lp="wide"``: the
    same arithmetic with the weight broadcast through LDS -- eight-wave workgroups over adjacent 64-pixel tiles x K slices, for batches that
    fill the chip that way (msm_conv1x1_in_multi_wide)).  xs: list of (B, Cin_l, H_l, W_l) NCHW maps (deepest Cin first),
    ws_packed / biases: per level (a bias may be None), out: (B, sum H_l*W_l, 64) token buffer or a token-range view of a larger
    one (level l fills its token range), stats: (L, B, 64, 2) float64 moments (accumulated into when ``stats_cleared``)."""
    L = len(xs)
    B = xs[0].shape[0]
    S = sum(x.shape[2] * x.shape[3] for x in xs)
    _chk(out, "out"), _c(stats, "stats", torch.float64)
    if tuple(out.shape) != (B, S, 64) or tuple(stats.shape) != (L, B, 64, 2) or out.stride(2) != 1 or out.stride(1) != 64 \
            or (B > 1 and out.stride(0) < S * 64):
        raise RuntimeError("conv1x1_in_multi: out must be (B, sum HW, 64) with strides (>= sum HW * 64, 64, 1) and stats (L, B, 64, 2)")
    for x, w, b in zip(xs, ws_packed, biases):
        _c(x, "x"), _c(w, "w_packed", torch.bfloat16 if lp else torch.float32), _c(b, "bias")
        if x.shape[0] != B or w.numel() != (128 if lp else 64) * x.shape[1]:
            raise RuntimeError("conv1x1_in_multi: inconsistent level shapes")
    vp = ctypes.c_void_p * L
    ia = ctypes.c_int32 * L
    xa, wa = vp(*[x.data_ptr() for x in xs]), vp(*[w.data_ptr() for w in ws_packed])
    ba = vp(*[0 if b is None else b.data_ptr() for b in biases])
    cin, hw = ia(*[x.shape[1] for x in xs]), ia(*[x.shape[2] * x.shape[3] for x in xs])
    name = "msm_conv1x1_in_multi_wide" if lp == "wide" else "msm_conv1x1_in_multi_lp" if lp else "msm_conv1x1_in_multi_f32"
    fn = getattr(lib(), name)
    rc = fn(L, ctypes.cast(xa, ctypes.c_void_p), ctypes.cast(wa, ctypes.c_void_p), ctypes.cast(ba, ctypes.c_void_p),
            ctypes.cast(cin, ctypes.c_void_p), ctypes.cast(hw, ctypes.c_void_p), _p(out), out.stride(0) if B > 1 else S * 64, _p(stats),
            1 if stats_cleared else 0, B, _stream())
    check(rc, name)
    return out, stats


def is_token_major(x):
    """True for a (B, C, H, W) tensor stored [B][H*W][C] (torch channels_last, possibly with a larger batch stride),
    e.g. the NCHW-shaped views the pixel decoder returns over its token buffer."""
    if x.dim() != 4:
        return False
    B, C, H, W = x.shape
    return x.stride(1) == 1 and x.stride(3) == C and x.stride(2) == W * C and x.stride(0) >= H * W * C and C > 1


def dense_kv_constant(cmat, cmat_width):
    """The (H*W, N) matrix of a separable constant [(H row vectors | W column vectors), N] (cmat_width = W; 0: cmat itself)."""
    if not cmat_width:
        return cmat
    h = cmat.shape[0] - cmat_width
    return (cmat[:h, None, :] + cmat[None, h:, :]).reshape(h * cmat_width, cmat.shape[1]).contiguous()


def kv_project(x, w, cmat, cmat_width=0):
    """Folded K/V projection: x (B, 64, H, W), w (N, 64), cmat (H*W, N) -> (B, H*W, N).
    x is contiguous NCHW or token-major (is_token_major).  Maps with N in {256, 512} take the weight-stationary
    kernel (csrc/kv_proj.hip); small NCHW ones, where copying w into every CU's LDS costs more than it saves, and
    other shapes take the tiled GEMM.  ``cmat_width`` = W: the constant is separable, cmat = (H + W, N) holds H row vectors then
    W column vectors and token (y, x) gets row[y] + col[x] (include/msm_hip.h)."""
    _chk(x, "x"), _c(w, "w"), _c(cmat, "cmat")
    B, C, H, W = x.shape
    N = w.shape[0]
    if cmat_width and (cmat_width != W or tuple(cmat.shape) != (H + W, N)):
        raise RuntimeError(f"kv_project: a separable constant must be ({H + W}, N) with cmat_width = {W}")
    tokens = is_token_major(x) and not x.is_contiguous()
    if not tokens:
        _c(x, "x")
    if C != 64 or N not in (256, 512) or (not tokens and B * H * W < 8192):
        return conv1x1_nchw_to_tokens(x.contiguous(), w, dense_kv_constant(cmat, cmat_width))
    if tuple(w.shape) != (N, C) or (not cmat_width and tuple(cmat.shape) != (H * W, N)):
        raise RuntimeError(f"kv_project: w must be (N, {C}) and cmat ({H * W}, N)")
    out = torch.empty((B, H * W, N), device=x.device, dtype=torch.float32)
    rc = lib().msm_kv_project_f32(_p(x), _p(w), _p(cmat), _p(out), B, C, H * W, N, 1 if tokens else 0, x.stride(0), int(cmat_width), _stream())
    check(rc, "msm_kv_project_f32")
    return out


def conv1x1_tokens_to_nchw(t, w, bias=None):
    """tokens (B, HW, Cin) -> (B, Cout, HW) with the weight as the MFMA A operand so that the
    NCHW output rows are written contiguously (bias is per output row)."""
    _c(t, "t"), _c(w, "w"), _c(bias, "bias")
    B, HW, Cin = t.shape
    Cout = w.shape[0]
    out = torch.empty((B, Cout, HW), device=t.device, dtype=torch.float32)
    # A = w (M=Cout, shared over batch), "W" = tokens of image b ([N=HW][K=Cin])
    rc = lib().msm_gemm_f32(_p(w), None, _p(t), _p(bias), _p(out), Cout, HW, Cin, B,
                            Cin, 1, 0, 0, HW * Cin, HW, 1, Cout * HW, 0,
                            0, 0, 0, 0, 2 if bias is not None else 0, 0, 1, _stream())
    check(rc, "msm_gemm_f32(conv1x1 to nchw)")
    return out


def conv3x3_tokens(t, w_tap_major, H, W):
    """3x3 / pad 1 convolution over an NHWC token map t (B, H*W, Cin) with weights permuted to
    (Cout, 9*Cin) tap-major; returns (B, H*W, Cout).  No bias (the reference layer has a norm)."""
    _c(t, "t"), _c(w_tap_major, "w")
    B, HW, Cin = t.shape
    Cout = w_tap_major.shape[0]
    out = torch.empty((B, HW, Cout), device=t.device, dtype=torch.float32)
    rc = lib().msm_gemm_f32(_p(t), None, _p(w_tap_major), None, _p(out), HW, Cout, 9 * Cin, B,
                            Cin, 1, HW * Cin, 0, 0, Cout, 1, HW * Cout, 0,
                            2, H, W, Cin, 0, 0, 1, _stream())
    check(rc, "msm_gemm_f32(conv3x3)")
    return out


def conv3x3_c64(t, w_tap_major, H, W, *, stats=None, stats_cleared=False, bf16=False, split=False):
    """3x3 / pad 1 convolution of a 64-channel token map t (B, H*W, 64) to 64 channels, weight (64, 9*64) tap-major, with
    the GroupNorm moments of the result as a by-product: returns (out (B, H*W, 64), stats (B, 64, 2) float64).  ``stats``
    given: accumulated into when ``stats_cleared`` (the caller zeroed it), else zeroed first.  ``bf16``: the low-precision
    mode (bf16 MFMA operands -- weight single, activations hi + lo --, fp32 accumulation and output); ``bf16="f16"``: IEEE-half
    operands, one term each (precision "f16").  ``split``: t is the
    (3, B, H*W, 64) bf16 planes of groupnorm_tokens(split_planes=True); fp32-accurate results from six bf16 MFMAs per product.
    A float16 ``t`` (groupnorm_tokens(out_f16=True)) is the "f16" form on operands already rounded: the same output bits, the kernel
    that keeps a unit's loads in flight at once (msm_conv3x3_c64_f16h)."""
    half_in = t.dtype == torch.float16
    if half_in and (split or bf16 != "f16"):
        raise RuntimeError("conv3x3_c64: a float16 map is the input of the bf16=\"f16\" form only")
    if split:
        _c(t, "t", torch.bfloat16)
        if t.dim() != 4 or t.shape[0] != 3:
            raise RuntimeError("conv3x3_c64(split=True) needs the (3, B, H*W, 64) bf16 planes")
        _, B, HW, C = t.shape
    else:
        _c(t, "t", torch.float16 if half_in else torch.float32)
        B, HW, C = t.shape
    _c(w_tap_major, "w"), _c(stats, "stats", torch.float64)
    if C != 64 or tuple(w_tap_major.shape) != (64, 576) or HW != H * W:
        raise RuntimeError("conv3x3_c64 needs a (B, H*W, 64) map and a (64, 576) tap-major weight")
    out = torch.empty((B, HW, C), device=t.device, dtype=torch.float32)
    if stats is None:
        stats = torch.empty((B, 64, 2), device=t.device, dtype=torch.float64)
        stats_cleared = False
    elif tuple(stats.shape) != (B, 64, 2):
        raise RuntimeError("stats must be (B, 64, 2) float64")
    name = "msm_conv3x3_c64_split" if split else "msm_conv3x3_c64_f16h" if half_in else ("msm_conv3x3_c64_f16" if bf16 == "f16" else "msm_conv3x3_c64_bf16" if bf16 else "msm_conv3x3_c64_f32")
    rc = getattr(lib(), name)(_p(t), _p(w_tap_major), _p(out), _p(stats), 1 if stats_cleared else 0, B, H, W, _stream())
    check(rc, name)
    return out, stats


def conv3x3_tokens_to_nchw(t, w_tap_major, bias, H, W, bf16=False):
    """3x3 / pad 1 convolution of an NHWC token map with the output written directly as NCHW (B, Cout, H*W)
    (SimpleBasePixelDecoder.mask_features, fpn.py:237-246: Conv2d 3x3 with bias).  64 input channels, Cout % 64 == 0 and
    W % 4 == 0 take the weight-stationary kernel (csrc/conv3x3.hip), other shapes the implicit GEMM.  ``bf16`` (low-precision
    mode, weight-stationary shapes only): the weight rounded to one bf16, activations as hi + lo operands, fp32 result;
    ``bf16="f16"``: IEEE-half operands, one term each."""
    _c(t, "t"), _c(w_tap_major, "w"), _c(bias, "bias")
    B, HW, Cin = t.shape
    Cout = w_tap_major.shape[0]
    out = torch.empty((B, Cout, HW), device=t.device, dtype=torch.float32)
    if Cin == 64 and Cout % 64 == 0 and Cout <= 1024 and W % 4 == 0 and HW == H * W:
        name = "msm_conv3x3_c64_nchw_f16" if bf16 == "f16" else "msm_conv3x3_c64_nchw_bf16" if bf16 else "msm_conv3x3_c64_nchw_f32"
        rc = getattr(lib(), name)(_p(t), _p(w_tap_major), _p(bias), _p(out), B, H, W, Cout, _stream())
        check(rc, name)
        return out
    rc = lib().msm_gemm_f32(_p(t), None, _p(w_tap_major), _p(bias), _p(out), HW, Cout, 9 * Cin, B,
                            Cin, 1, HW * Cin, 0, 0, 1, HW, Cout * HW, 0,
                            2, H, W, Cin, 1 if bias is not None else 0, 0, 1, _stream())
    check(rc, "msm_gemm_f32(conv3x3 -> nchw)")
    return out


def layernorm(x, g1, b1, *, parts=None, bias=None, l2norm=False, g2=None, b2=None, eps=1e-5):
    """LayerNorm(x + sum(parts) + bias) [-> unit length] [-> second LayerNorm].  Returns y or (y, y2)."""
    _c(x, "x"), _c(parts, "parts"), _c(bias, "bias"), _c(g1, "g1"), _c(b1, "b1"), _c(g2, "g2"), _c(b2, "b2")
    ref = x if x is not None else parts[0]
    E = ref.shape[-1]
    rows = ref.numel() // E
    y = torch.empty_like(ref)
    y2 = torch.empty_like(ref) if g2 is not None else None
    n_parts = 0 if parts is None else parts.shape[0]
    rc = lib().msm_layernorm_f32(_p(x), _p(parts), n_parts, rows * E, _p(bias), _p(g1), _p(b1),
                                 1 if l2norm else 0, _p(g2), _p(b2), _p(y), _p(y2), rows, E, eps, _stream())
    check(rc, "msm_layernorm_f32")
    return (y, y2) if g2 is not None else y


def groupnorm_tokens(x, gamma, beta, H, W, groups=32, *, up=None, up_hw=None, relu=False, eps=1e-5, stats=None, stats_ready=False,
                     split_planes=False, out_f16=False):
    """GroupNorm over an NHWC token map x (B, H*W, C); optionally adds the bilinear upsample of `up` (B, uh*uw, C) -- dense
    or a token-range slice of a larger buffer (row stride C, any batch stride) -- and applies ReLU.  ``stats``: a zeroed
    (B, C, 2) float64 scratch to accumulate the moments in (saves the fill launch) -- or, with ``stats_ready``, the finished
    moments of x (the producer of x accumulated them: no moments pass).  ``split_planes``: the result as three bf16 planes
    (3, B, H*W, C) with y = h + m + l exactly (the activation operand of conv3x3_c64(split=...)) instead of fp32.  ``out_f16``: the
    result as (B, H*W, C) float16, clamped to the half range (the operand conv3x3_c64(bf16="f16") rounds to, written once)."""
    if split_planes and out_f16:
        raise RuntimeError("groupnorm_tokens: split_planes and out_f16 are different output forms")
    _c(x, "x"), _c(gamma, "gamma"), _c(beta, "beta"), _chk(up, "up")
    B, HW, C = x.shape
    if stats_ready:
        _c(stats, "stats", torch.float64)
        if stats is None or tuple(stats.shape) != (B, C, 2):
            raise RuntimeError("stats_ready needs stats (B, C, 2) float64")
    else:
        stats = groupnorm_stats(x, stats)
    y = torch.empty((3,) + tuple(x.shape), device=x.device, dtype=torch.bfloat16) if split_planes else \
        torch.empty(x.shape, device=x.device, dtype=torch.float16) if out_f16 else torch.empty_like(x)
    uh, uw = (0, 0) if up is None else up_hw
    usb = 0
    if up is not None:
        if tuple(up.shape) != (B, uh * uw, C) or up.stride(2) != 1 or up.stride(1) != C or (B > 1 and up.stride(0) < uh * uw * C):
            raise RuntimeError("up must be (B, uh*uw, C) with strides (>= uh*uw*C, C, 1)")
        usb = up.stride(0) if B > 1 else 0
    name = "msm_groupnorm_apply_split" if split_planes else "msm_groupnorm_apply_f16" if out_f16 else "msm_groupnorm_apply_f32"
    rc = getattr(lib(), name)(_p(x), _p(stats), _p(gamma), _p(beta), _p(up), uh, uw, usb, _p(y), B, H, W, C, groups, eps, 1 if relu else 0, _stream())
    check(rc, name)
    return y


def groupnorm_nchw(x, stats, gamma, beta, *, groups=32, eps=1e-5, relu=False):
    """GroupNorm (+ReLU) of a token map x (B, HW, C) from its moments ``stats`` (B, C, 2) float64, written as NCHW planes
    (B, C, HW): the activation the folded mask step contracts with.  C <= 128, HW % 4 == 0."""
    _c(x, "x"), _c(stats, "stats", torch.float64), _c(gamma, "gamma"), _c(beta, "beta")
    B, HW, C = x.shape
    y = torch.empty((B, C, HW), device=x.device, dtype=torch.float32)
    rc = lib().msm_groupnorm_apply_nchw_f32(_p(x), _p(stats), _p(gamma), _p(beta), _p(y), B, HW, C, int(groups), float(eps),
                                            1 if relu else 0, _stream())
    check(rc, "msm_groupnorm_apply_nchw_f32")
    return y


def groupnorm_stats(x, stats=None):
    """Per-(image, channel) double (sum, sum of squares) of a token map x (B, HW, C) -> (B, C, 2) float64.  ``stats``: a
    ZEROED (B, C, 2) float64 tensor to accumulate into (then no fill launch is issued)."""
    _c(x, "x"), _c(stats, "stats", torch.float64)
    B, HW, C = x.shape
    cleared = stats is not None
    if stats is None:
        stats = torch.empty((B, C, 2), device=x.device, dtype=torch.float64)
    elif tuple(stats.shape) != (B, C, 2):
        raise RuntimeError("stats must be (B, C, 2) float64")
    check(lib().msm_groupnorm_stats_f32(_p(x), _p(stats), 1 if cleared else 0, B, HW, C, _stream()), "msm_groupnorm_stats_f32")
    return stats


def tokens_proj_nchw(x, w, bias=None, *, gn=None, relu=False):
    """1x1 convolution from tokens x (B, HW, 64) to NCHW (B, N, HW) with an optional GroupNorm (+ReLU) applied to x on
    the fly: gn = (stats from groupnorm_stats(x), gamma, beta, groups, eps).  N in {256, 512}; other shapes use
    conv1x1_tokens_to_nchw on a materialised GroupNorm output."""
    _c(x, "x"), _c(w, "w"), _c(bias, "bias")
    B, HW, C = x.shape
    N = w.shape[0]
    stats = gamma = beta = None
    groups, eps = 1, 0.0
    if gn is not None:
        stats, gamma, beta, groups, eps = gn
        _c(stats, "stats", torch.float64), _c(gamma, "gamma"), _c(beta, "beta")
    out = torch.empty((B, N, HW), device=x.device, dtype=torch.float32)
    rc = lib().msm_tokens_proj_nchw_f32(_p(x), _p(w), _p(bias), _p(stats), _p(gamma), _p(beta), int(groups), float(eps),
                                        1 if relu else 0, _p(out), B, C, HW, N, _stream())
    check(rc, "msm_tokens_proj_nchw_f32")
    return out


def pos_embed_sine(H, W, num_pos_feats, device, *, layout="nchw", add_c=None, temperature=10000.0,
                   scale=6.283185307179586):
    """PositionEmbeddingSine(normalize=True) for one map: (2N, H, W) for layout 'nchw',
    (H*W, 2N) for 'tokens' (optionally with a per-channel vector added)."""
    C = 2 * num_pos_feats
    _c(add_c, "add_c")
    if layout == "nchw":
        out = torch.empty((C, H, W), device=device, dtype=torch.float32)
        s_c, s_p = H * W, 1
    else:
        out = torch.empty((H * W, C), device=device, dtype=torch.float32)
        s_c, s_p = 1, C
    rc = lib().msm_pos_embed_sine(_p(out), H, W, num_pos_feats, s_c, s_p, _p(add_c), temperature, scale, _stream())
    check(rc, "msm_pos_embed_sine")
    return out


def transpose_last2(x):
    """(B, R, C) -> (B, C, R)"""
    _c(x, "x")
    B, R, C = x.shape
    out = torch.empty((B, C, R), device=x.device, dtype=torch.float32)
    check(lib().msm_transpose_f32(_p(x), _p(out), B, R, C, _stream()), "msm_transpose_f32")
    return out


def l2_normalize_nchw(x, eps=1e-12):
    """F.normalize(x, p=2, dim=1) for an NCHW map (B, C, H, W) (msm_l2_normalize_nchw_f32)."""
    _c(x, "x")
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    check(lib().msm_l2_normalize_nchw_f32(_p(x), _p(y), B, C, H * W, float(eps), _stream()), "msm_l2_normalize_nchw_f32")
    return y


def msda_locations(offsets, logits, reference_points, spatial_shapes):
    """The glue of the general MSDeformAttn.forward (ms_deform_attn.py:101-109): offsets (N,Lq,M,L,P,2), logits (N,Lq,M,L*P),
    reference_points (N,Lq,L,2), spatial_shapes (L,2) int64 -> (sampling_locations (N,Lq,M,L,P,2), attention_weights (N,Lq,M,L,P))."""
    _c(offsets, "offsets"), _c(logits, "logits"), _c(reference_points, "reference_points"), _c(spatial_shapes, "spatial_shapes", torch.int64)
    N, Lq, M, L, P, _ = offsets.shape
    if tuple(logits.shape) != (N, Lq, M, L * P) or tuple(reference_points.shape) != (N, Lq, L, 2):
        raise RuntimeError("msda_locations: logits must be (N,Lq,M,L*P) and reference_points (N,Lq,L,2)")
    loc = torch.empty_like(offsets)
    attn = torch.empty((N, Lq, M, L, P), device=offsets.device, dtype=torch.float32)
    check(lib().msm_msda_locations(_p(offsets), _p(logits), _p(reference_points), _p(spatial_shapes), _p(loc), _p(attn), N * Lq, M, L, P,
                                   _stream()), "msm_msda_locations")
    return loc, attn


def pack_mask_features_bf16(mask_features, f16=False):
    """fp32 NCHW (B, C, H, W) -> the channel-quad packed bf16 layout (B, C/4, H*W, 4) (int16 bit patterns) the bf16 mask
    step streams; do it once per forward, the 10 mask steps of a decoder pass reuse it.  ``f16`` (precision "f16"): IEEE-half
    elements, returned as a torch.float16 tensor -- mask_logits(packed_bf16=...) picks the fp16 MFMAs by that dtype."""
    _c(mask_features, "mask_features")
    B, C, H, W = mask_features.shape
    out = torch.empty((B, C // 4, H * W, 4), device=mask_features.device, dtype=torch.float16 if f16 else torch.int16)
    name = "msm_pack_mask_features_f16" if f16 else "msm_pack_mask_features_bf16"
    rc = getattr(lib(), name)(_p(mask_features), _p(out), B, C, H * W, _stream())
    check(rc, name)
    return out


def pack_mask_features_split(mask_features):
    """fp32 NCHW (B, 64, H, W) -> the exact three-term bf16 split (B, 3, 8, H*W, 8) (int16 bit patterns) of the fp32-accurate
    mask step on the bf16 matrix pipe (msm_mask_logits_split_fwd); once per forward."""
    _c(mask_features, "mask_features")
    B, C, H, W = mask_features.shape
    out = torch.empty((B, 3, C // 8, H * W, 8), device=mask_features.device, dtype=torch.int16)
    check(lib().msm_pack_mask_features_split(_p(mask_features), _p(out), B, C, H * W, _stream()), "msm_pack_mask_features_split")
    return out


def pool_mask_taps(act, sizes, zero_rows=0):
    """The 64-channel factored mask features act (B, 64, H, W) reduced bilinearly (align_corners=False) to each (th, tw) of
    ``sizes`` (H / th = W / tw in {2, 4, 8}: the mean of the four centre taps of every cell) -> list of token-major
    (B, th*tw, 64) tensors.  One launch (csrc/attn_mask.hip).  ``zero_rows`` = Q > 0: the same launch also clears a (B, Q) int32
    buffer (the row flags of the first attn_mask_pooled call), returned as a second result."""
    _c(act, "act")
    B, C, H, W = act.shape
    if C != 64 or not 1 <= len(sizes) <= 4:
        raise RuntimeError("pool_mask_taps needs a (B, 64, H, W) activation and 1..4 target sizes")
    outs = [torch.empty((B, int(th) * int(tw), 64), device=act.device, dtype=torch.float32) for th, tw in sizes]
    n = len(sizes)
    ths = (ctypes.c_int32 * n)(*[int(s[0]) for s in sizes])
    tws = (ctypes.c_int32 * n)(*[int(s[1]) for s in sizes])
    ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    flags = torch.empty((B, int(zero_rows)), device=act.device, dtype=torch.int32) if zero_rows else None
    rc = lib().msm_pool_mask_taps(_p(act), B, H, W, n, ctypes.cast(ths, ctypes.c_void_p), ctypes.cast(tws, ctypes.c_void_p),
                                  ctypes.cast(ptrs, ctypes.c_void_p), _p(flags), B * int(zero_rows), _stream())
    check(rc, "msm_pool_mask_taps")
    return (outs, flags) if zero_rows else outs


def attn_mask_pooled(mask_embed, pooled, *, qbias=None, row_any=None, bits=False, f16=False):
    # f16: False = the fp32 MFMA chain, True = single IEEE-half operands, "x3" = hi + lo IEEE-half pairs (three terms: fp32-class logits)
    """The next layer's attention mask from the pooled activation (pool_mask_taps): attn (B, Q, T) uint8 =
    (einsum('bqc,btc->bqt', mask_embed, pooled) + qbias[b, q]) < 0 and row_any (B, Q) int32 (1 where a row keeps an unmasked
    key).  mask_embed: (B, Q, 64), contiguous or the leading 64 columns of a wider row-major buffer; qbias (B, Q), any uniform
    element stride; row_any: an already ZEROED buffer (saves the fill launch).  Equal to the attention-mask output of
    mask_logits(..., target_size) on the unpooled activation up to fp32 summation order.
    ``bits`` (T % 16 == 0): attn comes back bit-packed and blocked instead -- int16 (B, ceil(Q / 112), T / 16, 16, 8), the layout of
    attn_pack_mask_bits, what hypersphere_attention_fused_kv reads.  ``f16`` (16-bit plans): IEEE-half operands on the 16-bit matrix
    pipe (fp32 accumulation) instead of the fp32 MFMA chain."""
    _chk(mask_embed, "mask_embed"), _c(pooled, "pooled"), _chk(qbias, "qbias")
    B, Q, C = mask_embed.shape
    T = pooled.shape[1]
    if C != 64 or tuple(pooled.shape) != (B, T, 64):
        raise RuntimeError("attn_mask_pooled needs a (B, Q, 64) embedding and a (B, T, 64) pooled activation")
    if mask_embed.stride(2) != 1 or (B > 1 and mask_embed.stride(0) != Q * mask_embed.stride(1)) or mask_embed.stride(1) < C:
        raise RuntimeError("mask_embed must be (B,Q,C) with unit column stride and uniformly spaced rows")
    qb_ld = 0
    if qbias is not None:
        if tuple(qbias.shape) != (B, Q) or (B > 1 and qbias.stride(0) != Q * qbias.stride(1)):
            raise RuntimeError("qbias must be (B,Q) with uniformly spaced elements")
        qb_ld = qbias.stride(1)
    if bits:
        if T % 16:
            raise RuntimeError("attn_mask_pooled(bits=True) needs T % 16 == 0")
        attn = torch.empty((B, (Q + 111) // 112, T // 16, 16, 8), device=mask_embed.device, dtype=torch.int16)
    else:
        attn = torch.empty((B, Q, T), device=mask_embed.device, dtype=torch.uint8)
    cleared = row_any is not None
    if row_any is None:
        row_any = torch.empty((B, Q), device=mask_embed.device, dtype=torch.int32)
    else:
        _c(row_any, "row_any", torch.int32)
    rc = lib().msm_attn_mask_pooled(_p(mask_embed), mask_embed.stride(1), _p(qbias), qb_ld, _p(pooled), _p(attn), _p(row_any),
                                    1 if cleared else 0, (1 if bits else 0) | (4 if f16 == "x3" else (2 if f16 else 0)), B, Q, T, _stream())
    check(rc, "msm_attn_mask_pooled")
    return attn, row_any


def mask_logits(mask_embed, mask_features, *, want_mask=True, target_size=None, sparse=False, row_any=None, packed_bf16=None,
                qbias=None, packed_split=None):
    """einsum('bqc,bchw->bqhw') (+ qbias[b, q]) with the next layer's attention mask fused.
    Returns (mask (B,Q,H,W) or None, attn (B,Q,th*tw) uint8 or None, row_any (B,Q) int32 or None).
    mask_embed: (B,Q,C), contiguous or the leading C columns of a wider row-major buffer; qbias: (B,Q) per-query constant
    (any uniform element stride) -- together they serve the folded form of the step (modeling.FoldedMaskFeatures).
    row_any: an already ZEROED (B,Q) int32 buffer (dec_heads(zero_row_any=True) provides one) -- saves the fill launch.
    packed_bf16: pack_mask_features_bf16(mask_features) -> the step runs with bf16 operands / fp32 accumulation.
    packed_split: pack_mask_features_split(mask_features) (C = 64) -> fp32-accurate on the bf16 matrix pipe (exact 3-term splits)."""
    _chk(mask_embed, "mask_embed"), _c(mask_features, "mask_features"), _chk(qbias, "qbias")
    B, Q, C = mask_embed.shape
    if mask_embed.stride(2) != 1 or (B > 1 and mask_embed.stride(0) != Q * mask_embed.stride(1)) or mask_embed.stride(1) < C:
        raise RuntimeError("mask_embed must be (B,Q,C) with unit column stride and uniformly spaced rows")
    embed_ld = mask_embed.stride(1)
    qb_ld = 0
    if qbias is not None:
        if tuple(qbias.shape) != (B, Q) or (B > 1 and qbias.stride(0) != Q * qbias.stride(1)):
            raise RuntimeError("qbias must be (B,Q) with uniformly spaced elements")
        qb_ld = qbias.stride(1)
    _, Cf, H, W = mask_features.shape
    if Cf != C:
        raise RuntimeError(f"mask_embed has {C} columns, mask_features {Cf} channels")
    dev = mask_embed.device
    mask = torch.empty((B, Q, H, W), device=dev, dtype=torch.float32) if want_mask else None
    attn = None
    th = tw = 0
    flags = 1 if sparse else 0
    if target_size is not None:
        th, tw = int(target_size[0]), int(target_size[1])
        attn = torch.empty((B, Q, th * tw), device=dev, dtype=torch.uint8)
        if row_any is None:
            row_any = torch.empty((B, Q), device=dev, dtype=torch.int32)
        else:
            _c(row_any, "row_any", torch.int32)
            flags |= 2
    else:
        row_any = None
    if packed_split is not None:
        _c(packed_split, "packed_split", torch.int16)
        rc = lib().msm_mask_logits_split_fwd(_p(mask_embed), _p(packed_split), _p(mask), _p(attn), _p(row_any),
                                             B, Q, C, H, W, th, tw, flags, embed_ld, _p(qbias), qb_ld, _stream())
        check(rc, "msm_mask_logits_split_fwd")
        return mask, attn, row_any
    if packed_bf16 is not None:
        half = packed_bf16.dtype == torch.float16                 # pack_mask_features_bf16(..., f16=True): MSM_MASK_F16
        _c(packed_bf16, "packed_bf16", torch.float16 if half else torch.int16)
        rc = lib().msm_mask_logits_bf16_fwd(_p(mask_embed), _p(packed_bf16), _p(mask), _p(attn), _p(row_any),
                                            B, Q, C, H, W, th, tw, flags | (4 if half else 0), embed_ld, _p(qbias), qb_ld, _stream())
        check(rc, "msm_mask_logits_bf16_fwd")
    else:
        rc = lib().msm_mask_logits_fwd(_p(mask_embed), _p(mask_features), _p(mask), _p(attn), _p(row_any),
                                       B, Q, C, H, W, th, tw, flags, embed_ld, _p(qbias), qb_ld, _stream())
        check(rc, "msm_mask_logits_fwd")
    return mask, attn, row_any


def hypersphere_attention(q, k, v, heads, *, masked=None, row_any=None, kappa=KAPPA, low_precision=False, keys_f16=False):
    """q (B,Lq,E), k/v (B,S,E) already projected (last dim contiguous, may be column slices of a
    wider buffer); masked uint8 (B,Lq,S).  Returns (B,Lq,E).
    low_precision (or bf16 k / v): bf16 MFMA operands with fp32 accumulation (msm_hypersphere_attn_lp_fwd); k and v may then be
    torch.bfloat16 (as written by kv_project_multi(..., out_dtype=torch.bfloat16)) or float32.
    keys_f16 (precision "f16"): q^ / k^ enter the score MFMAs as IEEE halves; a 16-bit k then holds HALF bit patterns (the K columns
    of kv_project_multi(..., keys_f16=True): a torch.bfloat16-typed view whose bits are fp16), v stays bf16."""
    kv_bf16 = k.dtype == torch.bfloat16
    if kv_bf16 != (v.dtype == torch.bfloat16):
        raise RuntimeError("k and v must have the same dtype")
    _chk(q, "q")
    for t, n in ((k, "k"), (v, "v")):
        _chk(t, n, torch.bfloat16 if kv_bf16 else torch.float32)
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if t.stride(-1) != 1:
            raise RuntimeError(f"{n}: last dim must be contiguous")
    _c(masked, "masked", torch.uint8), _c(row_any, "row_any", torch.int32)
    B, Lq, E = q.shape
    S = k.shape[1]
    if E != heads * 32:
        raise RuntimeError("head_dim must be 32")
    out = torch.empty((B, Lq, E), device=q.device, dtype=torch.float32)
    need = lib().msm_hypersphere_attn_workspace(B, Lq, S, heads)
    ws = torch.empty((need,), device=q.device, dtype=torch.float32)
    if kv_bf16 or low_precision:
        fmt = (2 if kv_bf16 else 3) if keys_f16 else (1 if kv_bf16 else 0)
        rc = lib().msm_hypersphere_attn_lp_fwd(_p(q), _p(k), _p(v), fmt, _p(masked), _p(row_any), _p(out), B, Lq, S, heads,
                                               q.stride(1), q.stride(0), k.stride(1), k.stride(0), v.stride(1), v.stride(0),
                                               kappa, _p(ws), need, _stream())
        check(rc, "msm_hypersphere_attn_lp_fwd")
        return out
    rc = lib().msm_hypersphere_attn_fwd(_p(q), _p(k), _p(v), _p(masked), _p(row_any), _p(out), B, Lq, S, heads,
                                        q.stride(1), q.stride(0), k.stride(1), k.stride(0), v.stride(1), v.stride(0),
                                        kappa, _p(ws), need, _stream())
    check(rc, "msm_hypersphere_attn_fwd")
    return out


def attn_pack_kv_weights(w, heads):
    """[K rows | V rows] folded projection weight (2 * heads * 32, 64) fp32 -> the fp16 MFMA fragments of hypersphere_attention_fused_kv
    (msm_attn_pack_kv_weights): (heads, 8, 64, 8) float16."""
    _c(w, "w")
    if tuple(w.shape) != (2 * heads * 32, 64):
        raise RuntimeError("attn_pack_kv_weights: w must be (2 * heads * 32, 64)")
    out = torch.empty((heads, 8, 64, 8), device=w.device, dtype=torch.float16)
    check(lib().msm_attn_pack_kv_weights(_p(w), _p(out), int(heads), _stream()), "msm_attn_pack_kv_weights")
    return out


def tokens_f16(x):
    """A level feature (B, 64, H, W) -- NCHW or a token-major (channels_last) view -- as the (B, H*W, 64) float16 token matrix
    hypersphere_attention_fused_kv streams (one pass per forward: every layer of the decoder reads the same feature)."""
    _chk(x, "x")
    B, C, H, W = x.shape
    if is_token_major(x) and not x.is_contiguous():
        t = x.permute(0, 2, 3, 1)                                  # (B, H, W, C) view: rows contiguous inside an image
        n = H * W * C
        if t.stride(3) == 1 and t.stride(2) == C and t.stride(1) == W * C and n % 8 == 0 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and B <= 65535:
            # a level's token range of a wider buffer (the encoder's concatenated levels): converted in place, one launch
            out = torch.empty((B, H * W, C), device=x.device, dtype=torch.float16)
            check(lib().msm_f32_to_f16_rows(_p(t), _p(out), B, n, t.stride(0), _stream()), "msm_f32_to_f16_rows")
            return out
        return to_f16(t.reshape(B, H * W, C).contiguous())
    x = x.contiguous()
    if C != 64:
        return to_f16(transpose_last2(x.view(B, C, H * W)))
    out = torch.empty((B, H * W, C), device=x.device, dtype=torch.float16)
    check(lib().msm_nchw_to_tokens_f16(_p(x), _p(out), B, C, H * W, _stream()), "msm_nchw_to_tokens_f16")
    return out


MASK_CONV_K = 576            # 9 taps x 64 channels; column 576 of a folded filter row is the per-query constant
MASK_CONV_LD = 580           # row length of mask_conv_fold_weight's GEMM output (16-byte aligned rows)


def mask_conv_fold_weight(weight, bias=None):
    """Conv2d(64, Cm, 3, padding=1) weight (Cm, 64, 3, 3) [+ bias (Cm,)] -> the (580, Cm) matrix Wf with
    gemm(e, Wf)[b, q] = [F[b, q, 64 * (3 ky + kx) + c] = sum_o e[b, q, o] W[o, c, ky, kx] | e[b, q, :] . bias | 0 0 0]:
    the per-query 3x3 filters mask_conv3x3_folded convolves the 64-channel feature with (the convolution folded into the embedding)."""
    Cm, C, kh, kw = weight.shape
    if (C, kh, kw) != (64, 3, 3):
        raise RuntimeError("mask_conv_fold_weight: a (Cm, 64, 3, 3) convolution weight")
    wf = torch.zeros((MASK_CONV_LD, Cm), device=weight.device, dtype=torch.float32)
    wf[:MASK_CONV_K] = weight.detach().float().permute(2, 3, 1, 0).reshape(MASK_CONV_K, Cm)
    if bias is not None:
        wf[MASK_CONV_K] = bias.detach().float()
    return wf


def mask_conv3x3_folded(x_f16, F, size, *, bits=True, row_any=None):
    """The UCN path's mask step with the mask_features convolution folded into the embedding (msm_mask_conv3x3_folded; 16-bit plans):
    x_f16 = tokens_f16(level feature) (B, H*W, 64) float16; F (B, Q, >= 577) fp32 = gemm(e, mask_conv_fold_weight(w, b)); size = (H, W),
    W % 16 == 0, Q <= 112.  bits=True: returns (mask bits int16 (B, 1, S / 16, 16, 8) as hypersphere_attention_fused_kv reads them
    [bit = logit < 0], row_any int32 (B, Q)); ``row_any`` given: a buffer the caller already cleared (dec_heads zero_row_any).
    bits=False: returns the fp32 logits (B, Q, H, W)."""
    _c(x_f16, "x_f16", torch.float16), _c(F, "F"), _c(row_any, "row_any", torch.int32)
    B, Q, ldf = F.shape
    H, W = int(size[0]), int(size[1])
    S = H * W
    if tuple(x_f16.shape) != (B, S, 64) or F.stride(2) != 1 or F.stride(1) % 4 or F.stride(0) % 4 or ldf < MASK_CONV_K + 1:
        raise RuntimeError("mask_conv3x3_folded: x_f16 (B, H*W, 64) float16 and F (B, Q, >= 577) with 16-byte aligned rows")
    if W % 16 or Q > 112:
        raise RuntimeError("mask_conv3x3_folded: W % 16 == 0 and Q <= 112")
    if bits:
        out = torch.empty((B, 1, S // 16, 16, 8), device=F.device, dtype=torch.int16)
        cleared = row_any is not None
        if row_any is None:
            row_any = torch.empty((B, Q), device=F.device, dtype=torch.int32)
        elif tuple(row_any.shape) != (B, Q):
            raise RuntimeError("mask_conv3x3_folded: row_any must be (B, Q)")
        check(lib().msm_mask_conv3x3_folded(_p(x_f16), _p(F), F.stride(1), F.stride(0), _p(out), _p(row_any), 1 if cleared else 0, None,
                                            B, Q, H, W, _stream()), "msm_mask_conv3x3_folded")
        return out, row_any
    out = torch.empty((B, Q, H, W), device=F.device, dtype=torch.float32)
    check(lib().msm_mask_conv3x3_folded(_p(x_f16), _p(F), F.stride(1), F.stride(0), None, None, 0, _p(out), B, Q, H, W, _stream()),
          "msm_mask_conv3x3_folded")
    return out


def attn_pack_mask_bits(masked):
    """uint8 mask (B, Lq, S) (nonzero = masked), S % 16 == 0 -> the bit-packed, blocked form hypersphere_attention_fused_kv reads
    (msm_attn_pack_mask_bits): int16 (B, ceil(Lq / 112), S / 16, 16, 8), word [b, qc, kb, lj, m] bit k = masked[b, 112 qc + 16 m + lj, 16 kb + k]."""
    _c(masked, "masked", torch.uint8)
    B, Lq, S = masked.shape
    if S % 16:
        raise RuntimeError("attn_pack_mask_bits: S must be a multiple of 16")
    out = torch.empty((B, (Lq + 111) // 112, S // 16, 16, 8), device=masked.device, dtype=torch.int16)
    assert out.numel() * 2 == lib().msm_attn_mask_bits_bytes(B, Lq, S)
    check(lib().msm_attn_pack_mask_bits(_p(masked), _p(out), B, Lq, S, _stream()), "msm_attn_pack_mask_bits")
    return out


def hypersphere_attention_fused_kv(q, x_f16, w_packed, rowcol, col_v_t, size, heads, *, masked=None, row_any=None, kappa=KAPPA, keys_f16=False):
    """Cross attention over a long key sequence with the folded K/V projection inside the kernel (msm_hypersphere_attn_fused_kv_fwd;
    16-bit plans): q (B, Lq, E) projected queries; x_f16 = tokens_f16(level feature) (B, H*W, 64); w_packed = attn_pack_kv_weights(w);
    rowcol (H + W, 2E) the separable constants of kv_project(cmat_width=W); col_v_t (E, W) = rowcol[H:, E:].t(); size = (H, W), W % 16 == 0.
    masked: uint8 (B, Lq, S), packed here, or the int16 bit-packed form (attn_pack_mask_bits / attn_mask_pooled(bits=True)).
    keys_f16: q^ / k^ as IEEE halves (precision "f16") instead of bf16.  Returns (B, Lq, E)."""
    _chk(q, "q"), _c(x_f16, "x_f16", torch.float16), _c(w_packed, "w_packed", torch.float16), _c(rowcol, "rowcol"), _c(col_v_t, "col_v_t")
    prepacked = masked is not None and masked.dtype == torch.int16            # attn_mask_pooled(bits=True) / attn_pack_mask_bits output
    _c(masked, "masked", torch.int16 if prepacked else torch.uint8), _c(row_any, "row_any", torch.int32)
    if q.stride(-1) != 1:
        raise RuntimeError("q: last dim must be contiguous")
    B, Lq, E = q.shape
    H, W = int(size[0]), int(size[1])
    S = H * W
    mshape = (B, (Lq + 111) // 112, S // 16, 16, 8) if prepacked else (B, Lq, S)
    if E != heads * 32 or tuple(x_f16.shape) != (B, S, 64) or tuple(rowcol.shape) != (H + W, 2 * E) or tuple(col_v_t.shape) != (E, W) \
            or tuple(w_packed.shape) != (heads, 8, 64, 8) or (masked is not None and tuple(masked.shape) != mshape):
        raise RuntimeError("hypersphere_attention_fused_kv: inconsistent shapes")
    out = torch.empty((B, Lq, E), device=q.device, dtype=torch.float32)
    need = lib().msm_hypersphere_attn_workspace(B, Lq, S, heads)
    ws = torch.empty((need,), device=q.device, dtype=torch.float32)
    bits = None if masked is None else (masked if prepacked else attn_pack_mask_bits(masked))     # one 16-byte load per lane and key block in the kernel
    rc = lib().msm_hypersphere_attn_fused_kv_fwd(_p(q), _p(x_f16), _p(w_packed), _p(rowcol), _p(col_v_t), 2 if keys_f16 else 1, _p(bits), _p(row_any),
                                                 _p(out), B, Lq, H, W, heads, q.stride(1), q.stride(0), kappa, _p(ws), need, _stream())
    check(rc, "msm_hypersphere_attn_fused_kv_fwd")
    return out


def hypersphere_attention_backward(q, k, v, heads, grad_out, *, masked=None, row_any=None, kappa=KAPPA):
    """Gradient of hypersphere_attention: returns (grad_q (B,Lq,E), grad_k (B,S,E), grad_v (B,S,E)), contiguous."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, n)
        if t.stride(-1) != 1:
            raise RuntimeError(f"{n}: last dim must be contiguous")
    _c(grad_out, "grad_out"), _c(masked, "masked", torch.uint8), _c(row_any, "row_any", torch.int32)
    if masked is not None and row_any is None:
        # a fully masked row has a zero softmax denominator in the recomputation (NaN gradients); the forward resets such rows
        # through row_any (DEC:618), so the backward needs the same flags
        raise RuntimeError("hypersphere_attention_backward: row_any is required whenever masked is given")
    B, Lq, E = q.shape
    S = k.shape[1]
    if E != heads * 32 or tuple(grad_out.shape) != (B, Lq, E):
        raise RuntimeError("head_dim must be 32 and grad_out (B,Lq,E)")
    gq = torch.empty((B, Lq, E), device=q.device, dtype=torch.float32)
    gk = torch.empty((B, S, E), device=q.device, dtype=torch.float32)
    gv = torch.empty((B, S, E), device=q.device, dtype=torch.float32)
    need = lib().msm_hypersphere_attn_bwd_workspace(B, Lq, heads)
    ws = torch.empty((need,), device=q.device, dtype=torch.float32)
    rc = lib().msm_hypersphere_attn_bwd(_p(q), _p(k), _p(v), _p(masked), _p(row_any), _p(grad_out), _p(gq), _p(gk), _p(gv), B, Lq, S, heads,
                                        q.stride(1), q.stride(0), k.stride(1), k.stride(0), v.stride(1), v.stride(0), kappa, _p(ws), need, _stream())
    check(rc, "msm_hypersphere_attn_bwd")
    return gq, gk, gv


# ----------------------------------------------------------------------------------------------
# fused decoder-layer tails (csrc/dec_chain.hip)
# ----------------------------------------------------------------------------------------------
def dec_pack_weight(w):
    """(N, K) torch Linear weight -> the MFMA-fragment order the dec_* kernels stream (include/msm_hip.h)."""
    _c(w, "w")
    N, K = w.shape
    packed = torch.empty_like(w)
    rc = lib().msm_dec_pack_weight(_p(w), _p(packed), N, K, _stream())
    check(rc, "msm_dec_pack_weight")
    return packed


def dec_pack_weight_bf16(w):
    """(N, K) fp32 Linear weight -> bf16 in the fragment order of the low-precision dec_* kernels (msm_dec_pack_weight_bf16);
    the result (dtype torch.bfloat16, shape (N, K), NOT row-major) is what dec_post_cross / dec_post_self / dec_heads take as
    a weight in that mode: they pick the bf16 entry points by the weights' dtype."""
    _c(w, "w")
    N, K = w.shape
    packed = torch.empty((N, K), device=w.device, dtype=torch.bfloat16)
    rc = lib().msm_dec_pack_weight_bf16(_p(w), _p(packed), N, K, _stream())
    check(rc, "msm_dec_pack_weight_bf16")
    return packed


def dec_pack_weight_bf16x2(w):
    """(N, K) fp32 Linear weight -> hi + lo bf16 fragments [bf16(W) | bf16(W - bf16(W))] (msm_dec_pack_weight_bf16x2; dtype
    torch.bfloat16, shape (N, 2 K), NOT row-major): what the "bf16" plan's tails multiply by since round 6 -- the dec_* wrappers pick the
    _bf16x2 entry points by the mark this function leaves on the tensor."""
    _c(w, "w")
    N, K = w.shape
    packed = torch.empty((N, 2 * K), device=w.device, dtype=torch.bfloat16)
    rc = lib().msm_dec_pack_weight_bf16x2(_p(w), _p(packed), N, K, _stream())
    check(rc, "msm_dec_pack_weight_bf16x2")
    packed._msm_x2 = True
    return packed


def dec_pack_weight_f16(w):
    """(N, K) fp32 Linear weight -> IEEE half in the fragment order of the 16-bit dec_* kernels (msm_dec_pack_weight_f16; dtype
    torch.float16, shape (N, K), NOT row-major): precision "f16" -- the dec_* wrappers pick the _f16 entry points by this dtype."""
    _c(w, "w")
    N, K = w.shape
    packed = torch.empty((N, K), device=w.device, dtype=torch.float16)
    rc = lib().msm_dec_pack_weight_f16(_p(w), _p(packed), N, K, _stream())
    check(rc, "msm_dec_pack_weight_f16")
    return packed


_DEC_SUFFIX = {torch.float32: "", torch.bfloat16: "_bf16", torch.float16: "_f16"}


def _wdtype(*ws):
    """Common dtype of the packed weight matrices of a dec_* call (fp32, bf16 or fp16 fragments, never mixed)."""
    dts = {w.dtype for w in ws if w is not None}
    if len(dts) != 1 or next(iter(dts)) not in _DEC_SUFFIX:
        raise RuntimeError(f"packed weights must be all float32, all bfloat16 or all float16, got {sorted(map(str, dts))}")
    return next(iter(dts))


def _dec_suffix(*ws):
    """Entry-point suffix for the packed weights of a dec_* call: by dtype, and "_bf16x2" for hi + lo bf16 fragments (all or none)."""
    wd = _wdtype(*ws)
    x2 = {bool(getattr(w, "_msm_x2", False)) for w in ws if w is not None}
    if len(x2) != 1:
        raise RuntimeError("packed weights must be all hi + lo bf16 fragments (dec_pack_weight_bf16x2) or none")
    return "_bf16x2" if x2.pop() else _DEC_SUFFIX[wd]


def dec_post_cross(attn_out, res, query_pos, wo, bo, ln_g, ln_b, w_in, b_in, eps=1e-5):
    """Weight matrices of the three dec_* calls are dec_pack_weight() outputs.
    x = LN(res + attn_out wo^T + bo); qk = (x + query_pos) w_in[:2E]^T + b_in[:2E]; v = x w_in[2E:]^T + b_in[2E:].
    attn_out/res (B,Q,E); query_pos (Q,E).  Returns (x (B,Q,E), qk (B,Q,2E), v (B,Q,E))."""
    wd = _wdtype(wo, w_in)
    for t, n in ((attn_out, "attn_out"), (res, "res"), (query_pos, "query_pos"), (bo, "bo"), (ln_g, "ln_g"), (ln_b, "ln_b"), (b_in, "b_in")):
        _c(t, n)
    _c(wo, "wo", wd), _c(w_in, "w_in", wd)
    B, Q, E = attn_out.shape
    x = torch.empty_like(attn_out)
    qk = torch.empty((B, Q, 2 * E), device=attn_out.device, dtype=torch.float32)
    v = torch.empty_like(attn_out)
    fn = getattr(lib(), "msm_dec_post_cross" + _dec_suffix(wo, w_in))
    rc = fn(_p(attn_out), _p(res), _p(query_pos), _p(wo), _p(bo), _p(ln_g), _p(ln_b), _p(w_in), _p(b_in), _p(x), _p(qk), _p(v), B * Q, Q, E,
            eps, _stream())
    check(rc, "msm_dec_post_cross")
    return x, qk, v


def dec_post_self(attn_out, res, wo, bo, ln_g, ln_b, w1, b1, w2, n_parts=None, eps=1e-5):
    """x = LN(res + attn_out wo^T + bo); parts (n_parts, B, Q, E): partial sums over equal slices of the hidden
    dimension of linear2(relu(linear1(x))), without linear2's bias.  Default n_parts: about 200 workgroups."""
    wd = _wdtype(wo, w1, w2)
    for t, n in ((attn_out, "attn_out"), (res, "res"), (bo, "bo"), (ln_g, "ln_g"), (ln_b, "ln_b"), (b1, "b1")):
        _c(t, n)
    _c(wo, "wo", wd), _c(w1, "w1", wd), _c(w2, "w2", wd)
    B, Q, E = attn_out.shape
    F = w1.shape[0]
    x = torch.empty_like(attn_out)
    chunks = F // E
    if n_parts is None:
        tiles = (B * Q + 15) // 16
        n_parts = max(d for d in range(1, chunks + 1) if chunks % d == 0 and (d == 1 or tiles * d <= 256))
    parts = torch.empty((n_parts, B, Q, E), device=attn_out.device, dtype=torch.float32)
    fn = getattr(lib(), "msm_dec_post_self" + _dec_suffix(wo, w1, w2))
    rc = fn(_p(attn_out), _p(res), _p(wo), _p(bo), _p(ln_g), _p(ln_b), _p(w1), _p(b1), _p(w2), F, _p(x), _p(parts), n_parts, B * Q, E, eps,
            _stream())
    check(rc, "msm_dec_post_self")
    return x, parts


def dec_heads(x, dec_g, dec_b, mlp, *, parts=None, bias=None, ln_g=None, ln_b=None, l2norm=False, wq=None, bq=None,
              query_pos=None, want_out=True, want_d=False, zero_row_any=False, eps=1e-5):
    """t = x + sum(parts) + bias [-> LN] [-> unit length]; d = LN_dec(t); e = MLP3(d); q = (t + query_pos) wq^T + bq.
    mlp = [(w0,b0),(w1,b1),(w2,b2)].  Returns (out|None, d|None, e, q|None), plus a zeroed (B,Q) int32 row_any buffer
    for the following mask step when zero_row_any."""
    wd = _wdtype(wq, *[w for w, _ in mlp])
    for i, t in enumerate([x, parts, bias, ln_g, ln_b, dec_g, dec_b, bq, query_pos] + [b for _, b in mlp]):
        _c(t, f"dec_heads arg {i}")
    for i, t in enumerate([wq] + [w for w, _ in mlp]):
        _c(t, f"dec_heads weight {i}", wd)
    B, Q, E = x.shape
    out = torch.empty_like(x) if want_out else None
    d = torch.empty_like(x) if want_d else None
    e = torch.empty_like(x)
    q = torch.empty_like(x) if wq is not None else None
    ra = torch.empty((B, Q), device=x.device, dtype=torch.int32) if zero_row_any else None
    n_parts = 0 if parts is None else parts.shape[0]
    (m0w, m0b), (m1w, m1b), (m2w, m2b) = mlp
    fn = getattr(lib(), "msm_dec_heads" + _dec_suffix(wq, *[w for w, _ in mlp]))
    rc = fn(_p(x), _p(parts), n_parts, _p(bias), _p(ln_g), _p(ln_b), 1 if l2norm else 0, _p(dec_g), _p(dec_b), _p(m0w), _p(m0b), _p(m1w),
            _p(m1b), _p(m2w), _p(m2b), _p(wq), _p(bq), _p(query_pos), _p(out), _p(d), _p(e), _p(q), _p(ra), B * Q, Q, E, eps, _stream())
    check(rc, "msm_dec_heads")
    return (out, d, e, q, ra) if zero_row_any else (out, d, e, q)


def dec_heads_mask(x, dec_g, dec_b, mlp, pooled, row_any, *, qcol=64, bits=False, f16=False, parts=None, bias=None, ln_g=None, ln_b=None, l2norm=False,
                   wq=None, bq=None, query_pos=None, want_out=True, want_d=False, eps=1e-5):
    """dec_heads (16-bit weight fragments) with the next layer's attention mask at key resolution as the kernel's epilogue
    (msm_dec_heads_mask): the values of dec_heads(...) followed by attn_mask_pooled(e[..., :64], pooled, qbias=e[..., qcol],
    row_any=row_any, bits=bits, f16=f16), bit for bit, in one launch.  ``mlp[-1]`` must be the folded final layer ([e Wm | e.bm | ..]);
    ``row_any`` (B, Q) int32 must arrive ZEROED.  Returns (out|None, d|None, e, q|None, attn, row_any)."""
    wd = _wdtype(wq, *[w for w, _ in mlp])
    if wd not in (torch.bfloat16, torch.float16) or _dec_suffix(wq, *[w for w, _ in mlp]) == "_bf16x2":
        raise RuntimeError("dec_heads_mask needs single bf16 or fp16 weight fragments (the fp32 plan and the hi + lo bf16 form keep the two launches)")
    for i, t in enumerate([x, parts, bias, ln_g, ln_b, dec_g, dec_b, bq, query_pos, pooled] + [b for _, b in mlp]):
        _c(t, f"dec_heads_mask arg {i}")
    for i, t in enumerate([wq] + [w for w, _ in mlp]):
        _c(t, f"dec_heads_mask weight {i}", wd)
    _c(row_any, "row_any", torch.int32)
    B, Q, E = x.shape
    T = pooled.shape[1]
    if tuple(pooled.shape) != (B, T, 64) or tuple(row_any.shape) != (B, Q):
        raise RuntimeError("dec_heads_mask needs a (B, T, 64) pooled activation and a (B, Q) row_any buffer")
    out = torch.empty_like(x) if want_out else None
    d = torch.empty_like(x) if want_d else None
    e = torch.empty_like(x)
    q = torch.empty_like(x) if wq is not None else None
    if bits:
        if T % 16:
            raise RuntimeError("dec_heads_mask(bits=True) needs T % 16 == 0")
        attn = torch.empty((B, (Q + 111) // 112, T // 16, 16, 8), device=x.device, dtype=torch.int16)
    else:
        attn = torch.empty((B, Q, T), device=x.device, dtype=torch.uint8)
    n_parts = 0 if parts is None else parts.shape[0]
    (m0w, m0b), (m1w, m1b), (m2w, m2b) = mlp
    rc = lib().msm_dec_heads_mask(_p(x), _p(parts), n_parts, _p(bias), _p(ln_g), _p(ln_b), 1 if l2norm else 0, _p(dec_g), _p(dec_b), _p(m0w), _p(m0b),
                                  _p(m1w), _p(m1b), _p(m2w), _p(m2b), _p(wq), _p(bq), _p(query_pos), _p(out), _p(d), _p(e), _p(q), _p(pooled), T,
                                  int(qcol), _p(attn), _p(row_any), (1 if bits else 0) | (2 if wd == torch.float16 else 0) | (4 if f16 else 0), B * Q, Q, E, eps, _stream())
    check(rc, "msm_dec_heads_mask")
    return out, d, e, q, attn, row_any


def dec_set_prefetch(tensors):
    """The NEXT dec_post_cross / dec_post_self / dec_heads(_mask) call of this thread touches the storage of ``tensors`` (<= 6 contiguous
    device tensors: the packed weights of the launches behind it in the chain) from an extra row of workgroups, so that every XCD's L2
    holds them when those launches start (msm_dec_set_prefetch).  Speed only; an empty list clears a pending request."""
    ts = [t for t in tensors if t is not None and t.numel() > 0][:6]
    for t in ts:
        if not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("dec_set_prefetch needs contiguous device tensors")
    n = len(ts)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[t.data_ptr() for t in ts])
    nbytes = (ctypes.c_int64 * max(n, 1))(*[t.numel() * t.element_size() for t in ts])
    check(lib().msm_dec_set_prefetch(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(nbytes, ctypes.c_void_p), n), "msm_dec_set_prefetch")


def l2_prefetch(tensors):
    """Touch the storage of up to 8 device tensors per launch so that every XCD's L2 holds it (msm_l2_prefetch): issued on a side stream
    beside a kernel that leaves the fabric idle, ahead of the kernel that streams these bytes.  Speed only."""
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    for t in ts:
        if not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("l2_prefetch needs contiguous device tensors")
    for i in range(0, len(ts), 8):
        grp = ts[i:i + 8]
        ptrs = (ctypes.c_void_p * len(grp))(*[t.data_ptr() for t in grp])
        nbytes = (ctypes.c_int64 * len(grp))(*[t.numel() * t.element_size() for t in grp])
        check(lib().msm_l2_prefetch(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(nbytes, ctypes.c_void_p), len(grp), _stream()), "msm_l2_prefetch")


def _msda_dtype(value, others):
    """float32 or float64 like the reference's dispatch (ms_deform_attn_cuda.cu:69); every floating tensor the same."""
    dt = value.dtype
    if dt not in (torch.float32, torch.float64):
        raise RuntimeError(f"ms_deform_attn: value must be float32 or float64, got {dt}")
    for t, name in others:
        _c(t, name, dt)
    return dt


def ms_deform_attn(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """Reference-ABI core op: value (N,S,M,D), shapes (L,2) int64, start (L,) int64,
    loc (N,Lq,M,L,P,2), w (N,Lq,M,L,P) -> (N,Lq,M*D).  float32 or float64."""
    dt = _msda_dtype(value, ((value, "value"), (sampling_locations, "sampling_locations"),
                             (attention_weights, "attention_weights")))
    _c(spatial_shapes, "spatial_shapes", torch.int64), _c(level_start_index, "level_start_index", torch.int64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    out = torch.empty((N, Lq, M * D), device=value.device, dtype=dt)
    fn, what = ((lib().msm_msdeform_attn_fwd, "msm_msdeform_attn_fwd") if dt == torch.float32 else
                (lib().msm_msdeform_attn_fwd_f64, "msm_msdeform_attn_fwd_f64"))
    rc = fn(_p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_locations),
            _p(attention_weights), _p(out), N, S, M, D, L, Lq, P, _stream())
    check(rc, what)
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, grad_output):
    """Reference-ABI backward: returns (grad_value, grad_sampling_loc, grad_attn_weight).  float32 or float64."""
    dt = _msda_dtype(value, ((value, "value"), (sampling_locations, "sampling_locations"),
                             (attention_weights, "attention_weights"), (grad_output, "grad_output")))
    _c(spatial_shapes, "spatial_shapes", torch.int64), _c(level_start_index, "level_start_index", torch.int64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    if tuple(grad_output.shape) != (N, Lq, M * D):
        raise RuntimeError(f"grad_output must be {(N, Lq, M * D)}, got {tuple(grad_output.shape)}")
    gv = torch.empty_like(value)
    gl = torch.empty_like(sampling_locations)
    gw = torch.empty_like(attention_weights)
    fn, what = ((lib().msm_msdeform_attn_bwd, "msm_msdeform_attn_bwd") if dt == torch.float32 else
                (lib().msm_msdeform_attn_bwd_f64, "msm_msdeform_attn_bwd_f64"))
    rc = fn(_p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_locations),
            _p(attention_weights), _p(grad_output), _p(gv), _p(gl), _p(gw),
            N, S, M, D, L, Lq, P, _stream())
    check(rc, what)
    return gv, gl, gw


def kv_project_multi(xs, ws, cmats, out_dtype=torch.float32, split=False, cmat_widths=None, keys_f16=False):
    """kv_project for a list of jobs in one launch: xs[j] (B, 64, H_j, W_j) contiguous NCHW or token-major
    (is_token_major), ws[j] (N, 64), cmats[j] (H_j*W_j, N) -> list of (B, H_j*W_j, N).  N in {256, 512}, <= 16 jobs.
    out_dtype torch.bfloat16: low-precision mode (bf16 MFMAs: w rounded to bf16, x as a hi + lo pair; bf16 output).
    split: fp32 results on the bf16 matrix pipe (exact three-term splits, msm_kv_project_multi_split).
    cmat_widths[j] = W_j: separable constants, cmats[j] (H_j + W_j, N) (see kv_project); all jobs or none.
    keys_f16 (out_dtype bfloat16, N = 512; precision "f16"): IEEE-half operands, and the K columns [:, :, :256] of the result hold
    HALF bit patterns (the tensor stays typed bfloat16: only hypersphere_attention(..., keys_f16=True) should read them), V bf16."""
    if keys_f16 and (out_dtype != torch.bfloat16 or ws[0].shape[0] != 512):
        raise RuntimeError("kv_project_multi: keys_f16 goes with out_dtype=torch.bfloat16 and N = 512 ([K | V])")
    if split and out_dtype != torch.float32:
        raise RuntimeError("kv_project_multi: split is the fp32-accurate form (float32 output)")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("kv_project_multi: out_dtype must be float32 or bfloat16")
    n = len(xs)
    B, N = xs[0].shape[0], ws[0].shape[0]
    outs, tok, sb, hw = [], [], [], []
    cws = [int(v) for v in cmat_widths] if cmat_widths is not None else [0] * n
    for x, w, c, cw in zip(xs, ws, cmats, cws):
        _chk(x, "x"), _c(w, "w"), _c(c, "cmat")
        Bx, C, H, W = x.shape
        t = is_token_major(x) and not x.is_contiguous()
        if not t:
            _c(x, "x")
        if Bx != B or C != 64 or w.shape[0] != N or tuple(w.shape) != (N, 64) or cw not in (0, W) \
                or tuple(c.shape) != ((H + W, N) if cw else (H * W, N)):
            raise RuntimeError("kv_project_multi: inconsistent job shapes")
        tok.append(1 if t else 0)
        sb.append(x.stride(0) if t else C * H * W)
        hw.append(H * W)
        outs.append(torch.empty((B, H * W, N), device=x.device, dtype=out_dtype))
    vp = ctypes.c_void_p * n
    arr = lambda ts: ctypes.cast(vp(*[t.data_ptr() for t in ts]), ctypes.c_void_p)
    ia, la = (ctypes.c_int32 * n), (ctypes.c_int64 * n)
    fn = (lib().msm_kv_project_multi_split if split else lib().msm_kv_project_multi_f32) if out_dtype == torch.float32 else lib().msm_kv_project_multi_bf16
    extra = (int(bool(keys_f16)),) if out_dtype == torch.bfloat16 else ()
    rc = fn(n, arr(xs), arr(ws), arr(cmats), arr(outs), ctypes.cast(ia(*hw), ctypes.c_void_p), ctypes.cast(ia(*tok), ctypes.c_void_p),
            ctypes.cast(la(*sb), ctypes.c_void_p), ctypes.cast(ia(*cws), ctypes.c_void_p), B, 64, N, *extra, _stream())
    check(rc, "msm_kv_project_multi")
    return outs


def value_to_head_major(value, heads):
    """(N, S, C) token-major value -> (N, heads, S, C/heads), the layout ms_deform_attn_encoder gathers fastest from."""
    _c(value, "value")
    N, S, C = value.shape
    out = torch.empty((N, heads, S, C // heads), device=value.device, dtype=torch.float32)
    rc = lib().msm_value_to_head_major_f32(_p(value), _p(out), N, S, heads, C // heads, _stream())
    check(rc, "msm_value_to_head_major_f32")
    return out


def ms_deform_attn_encoder(value, spatial_shapes, level_start_index, proj, heads, n_points):
    """Encoder self-attention form: proj (N,S,heads*L*P*3) raw offsets+logits; value (N,S,C) token-major, or
    (N,heads,S,C/heads) head-major as written by encoder_block(value_heads=heads).  Returns (N,S,C)."""
    _c(value, "value"), _c(proj, "proj")
    _c(spatial_shapes, "spatial_shapes", torch.int64), _c(level_start_index, "level_start_index", torch.int64)
    L = spatial_shapes.shape[0]
    if value.dim() == 4:
        N, M, S, D = value.shape
        if M != heads:
            raise RuntimeError("head-major value must be (N, heads, S, C/heads)")
        out = torch.empty((N, S, M * D), device=value.device, dtype=torch.float32)
        rc = lib().msm_msdeform_attn_enc_hm_fwd(_p(value), _p(spatial_shapes), _p(level_start_index), _p(proj), _p(out),
                                                N, S, M, D, L, n_points, _stream())
        check(rc, "msm_msdeform_attn_enc_hm_fwd")
        return out
    N, S, C = value.shape
    out = torch.empty((N, S, C), device=value.device, dtype=torch.float32)
    rc = lib().msm_msdeform_attn_enc_fwd(_p(value), _p(spatial_shapes), _p(level_start_index), _p(proj), _p(out),
                                         N, S, heads, C // heads, L, n_points, _stream())
    check(rc, "msm_msdeform_attn_enc_fwd")
    return out


def pack_msda_proj(wp, bp, heads, n_levels, n_points):
    """[sampling_offsets ; attention_weights] weight (heads*L*P*3, 64) and bias -> the per-head fragment-order stream and bias
    table of ms_deform_attn_encoder_fused (msm_msda_pack_proj)."""
    _c(wp, "wp"), _c(bp, "bp")
    if tuple(wp.shape) != (heads * n_levels * n_points * 3, 64) or bp.numel() != wp.shape[0]:
        raise RuntimeError(f"pack_msda_proj: weight must be ({heads * n_levels * n_points * 3}, 64) with a bias per row")
    wpack = torch.empty(heads * 3 * 4 * 64 * 4, device=wp.device, dtype=torch.float32)
    bpack = torch.empty(heads * 48, device=wp.device, dtype=torch.float32)
    check(lib().msm_msda_pack_proj(_p(wp), _p(bp), _p(wpack), _p(bpack), heads, n_levels, n_points, _stream()), "msm_msda_pack_proj")
    return wpack, bpack


def ms_deform_attn_encoder_fused(value_hm, spatial_shapes, level_start_index, src, pos, wpack, bpack, n_points):
    """Encoder self-attention with the sampling projection computed in the kernel: value_hm (N,heads,S,8) head-major,
    src (N,S,64) the layer input, pos (S,64); wpack / bpack from pack_msda_proj.  Returns (N,S,64)."""
    _c(value_hm, "value_hm"), _c(src, "src"), _c(pos, "pos"), _c(wpack, "wpack"), _c(bpack, "bpack")
    _c(spatial_shapes, "spatial_shapes", torch.int64), _c(level_start_index, "level_start_index", torch.int64)
    N, M, S, D = value_hm.shape
    if tuple(src.shape) != (N, S, M * D) or tuple(pos.shape) != (S, M * D):
        raise RuntimeError("ms_deform_attn_encoder_fused: src must be (N,S,C) and pos (S,C) for a (N,heads,S,C/heads) value")
    out = torch.empty((N, S, M * D), device=src.device, dtype=torch.float32)
    rc = lib().msm_msdeform_attn_enc_fused_fwd(_p(value_hm), _p(spatial_shapes), _p(level_start_index), _p(src), _p(pos), _p(wpack),
                                               _p(bpack), _p(out), N, S, M, D, spatial_shapes.shape[0], n_points, _stream())
    check(rc, "msm_msdeform_attn_enc_fused_fwd")
    return out


# ----------------------------------------------------------------------------------------------
# mean shift (lib/utils/mean_shift.py)
# ----------------------------------------------------------------------------------------------
def ms_pack_bf16(X):
    """X (n,64) fp32 -> the bf16 copy (msm_ms_bf16_rows(n), 64) (rows zero-padded to whole 32-point slabs) that the "bf16" precision
    of the clustering streams instead of X (ms_select_seeds / ms_hill_climb with xb=...)."""
    _c(X, "X")
    n, d = X.shape
    xb = torch.empty((lib().msm_ms_bf16_rows(n), d), device=X.device, dtype=torch.bfloat16)
    check(lib().msm_ms_pack_bf16(_p(X), n, d, _p(xb), _stream()), "msm_ms_pack_bf16")
    return xb


def _check_xb(xb, n, d, who):
    """``xb`` must be the whole padded copy ms_pack_bf16 makes: the kernels read entire 32-row slabs, so a shorter buffer (even one
    with n rows) would be read past its end."""
    _c(xb, "xb", torch.bfloat16)
    want = (int(lib().msm_ms_bf16_rows(n)), d)
    if tuple(xb.shape) != want:
        raise RuntimeError(f"{who}: xb has shape {tuple(xb.shape)}, expected ms_pack_bf16(X) of shape {want}")
    return xb


def ms_select_seeds(X, num_seeds, first_index, stepwise=False, _test_give_up=False, xb=None):
    """Farthest-point seeding.  X (n,64) unit rows.  Returns (seeds (S,64), indices int64 (S,)).  The single-launch
    persistent kernel (maps up to 393 216 rows) needs its workgroups co-resident; if other work holds the CUs it gives up
    and every index is -1 -- callers re-issue with ``stepwise=True`` (mean_shift.mean_shift_smart_init does).
    ``xb`` (ms_pack_bf16(X); precision "bf16"): maps beyond the fp32 persistent kernel's reach work on the bf16 copy -- one
    persistent launch that keeps 917 504 rows on chip (VGPRs + LDS) and streams the rest per step, or (``stepwise``) one
    launch per step over the copy; distances are those of the rounded points, so the indices may differ from the fp32 path's."""
    _c(X, "X")
    n, d = X.shape
    seeds = torch.empty((num_seeds, d), device=X.device, dtype=torch.float32)
    idx = torch.empty((num_seeds,), device=X.device, dtype=torch.int64)
    need = lib().msm_ms_seed_workspace(n)
    ws = torch.empty((need,), device=X.device, dtype=torch.float32)
    if xb is not None and n > 393216:
        _check_xb(xb, n, d, "ms_select_seeds")
        rc = lib().msm_ms_select_seeds_bf16(_p(xb), _p(X), n, d, num_seeds, int(first_index), _p(seeds), _p(idx), _p(ws), need,
                                            (1 if stepwise else 0) | (2 if _test_give_up else 0), _stream())
        check(rc, "msm_ms_select_seeds_bf16")
        return seeds, idx
    rc = lib().msm_ms_select_seeds(_p(X), n, d, num_seeds, int(first_index), _p(seeds), _p(idx), _p(ws), need,
                                   (1 if stepwise else 0) | (2 if _test_give_up else 0), _stream())
    check(rc, "msm_ms_select_seeds")
    return seeds, idx


def ms_hill_climb(X, Z, kappa, iters, precision="f32", xb=None):
    """iters x { Z = normalize(exp(kappa Z X^T) X) }; returns the updated copy of Z.  precision "f32": fp32 MFMAs;
    "f32_split": fp32 results from six bf16 MFMAs per product on exact three-term splits (msm_ms_hill_climb_split);
    "bf16": single bf16 products over the bf16 copy ``xb`` = ms_pack_bf16(X) (made here when not given), seeds as h + l terms."""
    if precision not in ("f32", "f32_split", "bf16"):
        raise ValueError(f"ms_hill_climb: precision must be 'f32', 'f32_split' or 'bf16', not {precision!r}")
    _c(X, "X"), _c(Z, "Z")
    n, d = X.shape
    S = Z.shape[0]
    Z = Z.clone()
    if precision == "bf16":
        xb = ms_pack_bf16(X) if xb is None else _check_xb(xb, n, d, "ms_hill_climb")
        need = lib().msm_ms_hill_climb_workspace(n, S)
        ws = torch.empty((need,), device=X.device, dtype=torch.float32)
        check(lib().msm_ms_hill_climb_bf16(_p(xb), n, d, _p(Z), S, float(kappa), int(iters), _p(ws), need, _stream()), "msm_ms_hill_climb_bf16")
        return Z
    need = (lib().msm_ms_hill_climb_split_workspace if precision == "f32_split" else lib().msm_ms_hill_climb_workspace)(n, S)
    ws = torch.empty((need,), device=X.device, dtype=torch.float32)
    fn = lib().msm_ms_hill_climb_split if precision == "f32_split" else lib().msm_ms_hill_climb
    rc = fn(_p(X), n, d, _p(Z), S, float(kappa), int(iters), _p(ws), need, _stream())
    check(rc, "msm_ms_hill_climb_split" if precision == "f32_split" else "msm_ms_hill_climb")
    return Z


def ms_assign(X, Z, seed_labels, num_labels):
    """labels[i] = seed_labels[first argmin_s 0.5(1 - X_i.Z_s)], counts = bincount(labels)."""
    _c(X, "X"), _c(Z, "Z"), _c(seed_labels, "seed_labels", torch.int64)
    n, d = X.shape
    labels = torch.empty((n,), device=X.device, dtype=torch.int64)
    counts = torch.empty((num_labels,), device=X.device, dtype=torch.int64)
    rc = lib().msm_ms_assign(_p(X), n, d, _p(Z), Z.shape[0], _p(seed_labels), _p(labels), _p(counts), num_labels, _stream())
    check(rc, "msm_ms_assign")
    return labels, counts


def ms_connected_components(Z, epsilon):
    """mean_shift.py:41-76 on the device: Z (S,64) -> (seed_labels (S,) int64, num (2,) int32 = [labels that survive =
    len(unique(seed_labels)), labels created])."""
    _c(Z, "Z")
    S = Z.shape[0]
    seed_labels = torch.empty((S,), device=Z.device, dtype=torch.int64)
    num = torch.empty((2,), device=Z.device, dtype=torch.int32)
    check(lib().msm_ms_connected_components(_p(Z), S, Z.shape[1], float(epsilon), _p(seed_labels), _p(num), _stream()),
          "msm_ms_connected_components")
    return seed_labels, num


def ms_relabel_largest_zero(labels, counts, num_alive=None):
    """mean_shift.py:211-227 in place: label 0 <-> the first-argmax label of counts[:num], num = len(unique(seed_labels)) read from
    the device tensor ``num_alive`` (ms_connected_components()[1]) when given, else every entry of counts."""
    _c(labels, "labels", torch.int64), _c(counts, "counts", torch.int64), _c(num_alive, "num_alive", torch.int32)
    rc = lib().msm_ms_relabel_largest_zero(_p(labels), labels.numel(), _p(counts), counts.numel(), _p(num_alive), _stream())
    check(rc, "msm_ms_relabel_largest_zero")
    return labels


# ----------------------------------------------------------------------------------------------
# instance post-processing (pretrained_meanshiftformer_model.py:337-343, 461-497)
# ----------------------------------------------------------------------------------------------
def topk_class_scores(pred_logits, topk, gather=None, gather_cols=None):
    """Top-K (query, class) pairs of softmax(pred_logits)[..., :-1] per image (PM:461-470): (scores (B,T), classes (B,T) int64,
    query index (B,T) int32).  ``gather``: a (B, Q, >= gather_cols) per-query matrix with unit column stride; the selected rows'
    leading ``gather_cols`` columns come back as a fourth result (B, T, gather_cols), copied by the same launch."""
    _c(pred_logits, "pred_logits")
    B, Q, K1 = pred_logits.shape
    dev = pred_logits.device
    scores = torch.empty((B, topk), device=dev, dtype=torch.float32)
    classes = torch.empty((B, topk), device=dev, dtype=torch.int64)
    qidx = torch.empty((B, topk), device=dev, dtype=torch.int32)
    if gather is None:
        rc = lib().msm_topk_class_scores(_p(pred_logits), B, Q, K1, topk, _p(scores), _p(classes), _p(qidx), _stream())
        check(rc, "msm_topk_class_scores")
        return scores, classes, qidx
    _chk(gather, "gather")
    cols = int(gather_cols)
    if gather.dim() != 3 or gather.shape[0] != B or gather.shape[1] != Q or gather.shape[2] < cols or gather.stride(2) != 1 \
            or (B > 1 and gather.stride(0) != Q * gather.stride(1)):
        raise RuntimeError("gather must be (B, Q, >= gather_cols) with unit column stride and uniformly spaced rows")
    sel = torch.empty((B, topk, cols), device=dev, dtype=torch.float32)
    rc = lib().msm_topk_class_scores_gather(_p(pred_logits), B, Q, K1, topk, _p(scores), _p(classes), _p(qidx), _p(gather),
                                            gather.stride(1), cols, _p(sel), _stream())
    check(rc, "msm_topk_class_scores_gather")
    return scores, classes, qidx, sel


def instance_postprocess(mask_logits, query_index, image_size, class_scores=None, padded_size=None):
    """mask_logits (B,Q,h,w), query_index int32 (B,T) -> (pred_masks (B,T,H,W) float 0/1,
    score (B,T) = mean mask probability [* class_scores], boxes (B,T,4)).  The logits are upsampled to ``padded_size``
    (the frame the network saw, default = image_size) and cropped to image_size = (H, W), as the reference does for
    inputs padded to the size divisibility (PM:337-343 + sem_seg_postprocess, PM:354-357)."""
    _c(mask_logits, "mask_logits"), _c(query_index, "query_index", torch.int32), _c(class_scores, "class_scores")
    B, Q, h, w = mask_logits.shape
    T = query_index.shape[1]
    H, W = int(image_size[0]), int(image_size[1])
    Hs, Ws = (H, W) if padded_size is None else (int(padded_size[0]), int(padded_size[1]))
    dev = mask_logits.device
    masks = torch.empty((B, T, H, W), device=dev, dtype=torch.float32)
    score = torch.empty((B, T), device=dev, dtype=torch.float32)
    boxes = torch.empty((B, T, 4), device=dev, dtype=torch.float32)
    ws = torch.empty((int(lib().msm_instance_postprocess_workspace(B, T, H, W)),), device=dev, dtype=torch.float32)
    rc = lib().msm_instance_postprocess(_p(mask_logits), _p(query_index), _p(class_scores), _p(masks), _p(score), _p(boxes),
                                        B, Q, T, h, w, H, W, Hs, Ws, _p(ws), _stream())
    check(rc, "msm_instance_postprocess")
    return masks, score, boxes


def pack_encoder_prologue(wv, wp):
    """Weight stream of msm_encoder_prologue_fwd: value_proj (64,64) then [sampling_offsets | attention_weights]
    (proj_width,64) as consecutive 16-row blocks, zero-padded to the stream length."""
    pw = wp.shape[0]
    n = int(lib().msm_encoder_prologue_stream_floats(pw))
    out = torch.zeros(n, device=wv.device, dtype=torch.float32)
    out[:64 * 64] = wv.reshape(-1)
    out[64 * 64:64 * 64 + pw * 64] = wp.reshape(-1)
    return out


def encoder_prologue(raw, stats, gn_params, level_starts, stream, small, pos, proj_width, *, groups=32, eps=1e-5, value_heads=0,
                     bf16_hm=False):
    """raw (B,S,64) concatenated input projections (conv1x1_in), stats (L,B,64,2) float64 their GroupNorm moments,
    gn_params (L,2,64) = gamma, beta, level_starts: L+1 token offsets (0..S).  Normalises raw IN PLACE (-> src) and
    returns (src, value, proj) for the first encoder layer: value (B,S,64) or head-major (B,heads,S,64/heads),
    proj (B,S,proj_width) = [sampling_offsets | attention_weights](src + pos).  bf16_hm (8 heads, proj_width 288): value and
    proj are the bf16 plan's head-major fp16 tensors (B,8,S,8) and (B,8,S,36) (encoder_block_hm)."""
    _c(raw, "raw"), _c(stats, "stats", torch.float64), _c(gn_params, "gn_params"), _c(stream, "stream"), _c(small, "small"), _c(pos, "pos")
    B, S, C = raw.shape
    L = len(level_starts) - 1
    if C != 64 or tuple(stats.shape) != (L, B, 64, 2) or tuple(gn_params.shape) != (L, 2, 64) or tuple(pos.shape) != (S, 64):
        raise RuntimeError("encoder_prologue: inconsistent shapes")
    if small.numel() != 64 + proj_width:
        raise RuntimeError("encoder_prologue: small must hold the 64 value_proj biases and the proj_width projection biases")
    dev = raw.device
    if bf16_hm:
        value = torch.empty((B, 8, S, 8), device=dev, dtype=torch.float16)
        proj = torch.empty((B, 8, S, PROJ_REC_FLOATS), device=dev, dtype=torch.float32)
    else:
        value = torch.empty((B, value_heads, S, 64 // value_heads) if value_heads else (B, S, 64), device=dev, dtype=torch.float32)
        proj = torch.empty((B, S, proj_width), device=dev, dtype=torch.float32) if proj_width else None     # 0: value projection only
    ls = (ctypes.c_int32 * (L + 1))(*[int(v) for v in level_starts])
    rc = lib().msm_encoder_prologue_fwd(_p(raw), _p(stats), _p(gn_params), ctypes.cast(ls, ctypes.c_void_p), L, int(groups), float(eps),
                                        _p(stream), _p(small), _p(pos), _p(raw), _p(value), _p(proj), B, S, int(proj_width),
                                        int(value_heads), int(bool(bf16_hm)), _stream())
    check(rc, "msm_encoder_prologue_fwd")
    return raw, value, proj


def label_stats(labels, weight=None, k=1024):
    """labels (B,H,W) float32 with integer values in [0,k), weight (B,H,W) float32 or None ->
    (stats (B,k,5) int32 = area, x_min, y_min, x_max, y_max; wsum (B,k) float32; overflow (B,) int32)."""
    _c(labels, "labels"), _c(weight, "weight")
    B, H, W = labels.shape
    dev = labels.device
    stats = torch.empty((B, k, 5), device=dev, dtype=torch.int32)
    wsum = torch.empty((B, k), device=dev, dtype=torch.float32)
    overflow = torch.empty((B,), device=dev, dtype=torch.int32)
    rc = lib().msm_label_stats(_p(labels), _p(weight), _p(stats), _p(wsum), _p(overflow), B, H, W, int(k), _stream())
    check(rc, "msm_label_stats")
    return stats, wsum, overflow


def label_image(masks, inst_labels):
    """masks (B,K,H,W) float (non-zero = inside), inst_labels (B,K) float -> (B,H,W) float label images (msm_label_image)."""
    _c(masks, "masks"), _c(inst_labels, "inst_labels")
    B, K, H, W = masks.shape
    out = torch.empty((B, H, W), device=masks.device, dtype=torch.float32)
    check(lib().msm_label_image(_p(masks), _p(inst_labels), _p(out), B, K, H, W, _stream()), "msm_label_image")
    return out


def crop_resize(rgb, depth, labels, table, size):
    """ROI crops of a batch of frames in one launch (msm_crop_resize): rgb / depth (F,3,H,W), labels (F,H,W) float, table (N,8)
    int32 rows (frame, label, x0, y0, x1, y1, 0, 0) -> (rgb_crops (N,3,S,S), mask_crops (N,S,S), depth_crops or None)."""
    _c(rgb, "rgb"), _c(depth, "depth"), _c(labels, "labels"), _c(table, "table", torch.int32)
    N = table.shape[0]
    F_, _, H, W = rgb.shape
    dev = rgb.device
    rgb_out = torch.empty((N, 3, size, size), device=dev, dtype=torch.float32)
    mask_out = torch.empty((N, size, size), device=dev, dtype=torch.float32)
    depth_out = torch.empty((N, 3, size, size), device=dev, dtype=torch.float32) if depth is not None else None
    check(lib().msm_crop_resize(_p(rgb), _p(depth), _p(labels), _p(table), _p(rgb_out), _p(depth_out), _p(mask_out), N, H, W, int(size),
                                _stream()), "msm_crop_resize")
    return rgb_out, mask_out, depth_out


def paste_labels(renum, table, order, frame_start, frames, H, W):
    """Paste-back of a batch (msm_paste_labels): renum (N,S,S) float, table (N,8) int32, order (N,) int32, frame_start (F+1,)
    int32 -> refined (F,H,W) float."""
    _c(renum, "renum"), _c(table, "table", torch.int32), _c(order, "order", torch.int32), _c(frame_start, "frame_start", torch.int32)
    refined = torch.empty((frames, H, W), device=renum.device, dtype=torch.float32)
    check(lib().msm_paste_labels(_p(renum), _p(table), _p(order), _p(frame_start), _p(refined), frames, H, W, renum.shape[-1], _stream()),
          "msm_paste_labels")
    return refined


# ----------------------------------------------------------------------------------------------
# fused encoder block (msdeformattn.py:122-131)
# ----------------------------------------------------------------------------------------------
def pack_encoder_block(wo, w1, w2, wv=None, wp=None):
    """Pack one encoder layer's matrices into the weight stream consumed by msm_encoder_block_fwd.

    Stream = chunks of 8 blocks, one block = 1024 floats (4 KiB):
      chunk 0            : output_proj  -- 4 row blocks [16 out rows][64 k] (+4 zero blocks)
      chunks 1..d_ffn/64 : 4 x ( linear1 row block [16 hidden rows][64 k] , linear2 block [64 out rows][16 hidden] )
      then (only with the next layer's wv/wp): value_proj 4 row blocks, then [offsets|weights] row blocks,
      continuing into further chunks of 8.
    A "row block" is 16 consecutive rows of a (N, 64) weight; the kernel applies the LDS swizzle itself."""
    dev = wo.device
    d_ffn = w1.shape[0]
    blocks = [wo.reshape(4, 1024)] + [torch.zeros(4, 1024, device=dev)]
    w1b = w1.reshape(d_ffn // 16, 1024)                                                # (hb, 16 rows * 64 k)
    w2b = w2.reshape(64, d_ffn // 16, 16).permute(1, 0, 2).reshape(d_ffn // 16, 1024)    # (hb, 64 rows * 16 k)
    blocks.append(torch.stack([w1b, w2b], 1).reshape(-1, 1024))                        # interleaved per hb
    if wv is not None:
        npb = wp.shape[0] // 16
        tail = torch.cat([wv.reshape(4, 1024), wp.reshape(npb, 1024)], 0)
        pad = (-tail.shape[0]) % 8
        blocks += [tail, torch.zeros(pad, 1024, device=dev)]
    return torch.cat(blocks, 0).reshape(-1).contiguous()


def pack_encoder_block_split(wo, w1, w2, wv=None, wp=None):
    """One encoder layer's matrices as the triple-split weight stream of msm_encoder_block_split_fwd (include/msm_hip.h):
    every fp32 weight as w = h + m + l with h = bf16(w), m = bf16(w - h), l = bf16(w - h - m); 2-KiB blocks in the fragment
    order of v_mfma_f32_16x16x32_bf16, a logical block = its (h, m, l) blocks, 12 blocks per stage.
    Returns an int16 tensor (bf16 bit patterns)."""
    dev = wo.device
    d_ffn = w1.shape[0]

    def rowblocks(w):       # (N, 64) -> (N/16, 1024): block[G][lq][lj][hh][c] = W[r0 + lj][(2G + hh)*16 + lq*4 + c]
        return w.reshape(-1, 16, 2, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(-1, 1024)

    def w2pairs(w):         # (64, d_ffn) -> (d_ffn/32, 2048): [ob][lq][lj][hh][c] = W[ob*16 + lj][(2P + hh)*16 + lq*4 + c]
        return w.reshape(4, 16, d_ffn // 32, 2, 4, 4).permute(2, 0, 4, 1, 3, 5).reshape(-1, 2048)

    def split3(w):
        h = w.to(torch.bfloat16).float()
        m = (w - h).to(torch.bfloat16).float()
        return h, m, (w - h - m).to(torch.bfloat16).float()

    def triples(parts):                                   # three (n, k) lists -> (n, 3k) as [h | m | l] per logical block
        return torch.cat(list(parts), 1)

    w1t = triples(rowblocks(t) for t in split3(w1)).reshape(d_ffn // 32, 2 * 3 * 1024)        # per stage: W1(q0) h,m,l | W1(q1) h,m,l
    w2t = triples(w2pairs(t) for t in split3(w2))                                            # per stage: W2 h | m | l (4 KiB each)
    blocks = [triples(rowblocks(t) for t in split3(wo)).reshape(-1), torch.cat([w1t, w2t], 1).reshape(-1)]
    if wv is not None:
        blocks.append(triples(rowblocks(t) for t in split3(wv)).reshape(-1))
        pt = triples(rowblocks(t) for t in split3(wp)).reshape(-1)
        blocks += [pt, torch.zeros((-(pt.numel() // 1024)) % 12 * 1024, device=dev)]
    out = torch.cat(blocks, 0).to(torch.bfloat16).contiguous().view(torch.int16).reshape(-1)
    need = int(lib().msm_encoder_block_split_stream_bytes(d_ffn, 0 if wp is None else wp.shape[0]))
    if out.numel() * 2 != need:
        raise RuntimeError(f"pack_encoder_block_split: built {out.numel() * 2} bytes, the kernel expects {need}")
    return out


def pack_encoder_block_lp(wo, w1, w2, wv=None, wp=None):
    """One encoder layer's matrices as the weight stream of msm_encoder_block_lp_fwd (include/msm_hip.h): the low-precision
    mode on the K = 32 kernel -- projections as [h, m] bf16 pairs, linear1 / linear2 as single bf16 copies, three hidden pairs
    per 12-block stage.  Returns an int16 tensor (bf16 bit patterns)."""
    dev = wo.device
    d_ffn = w1.shape[0]

    def rowblocks(w):       # (N, 64) -> (N/16, 1024): block[G][lq][lj][hh][c] = W[r0 + lj][(2G + hh)*16 + lq*4 + c]
        return w.reshape(-1, 16, 2, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(-1, 1024)

    def w2pairs(w):         # (64, d_ffn) -> (d_ffn/32, 2048): [ob][lq][lj][hh][c] = W[ob*16 + lj][(2P + hh)*16 + lq*4 + c]
        return w.reshape(4, 16, d_ffn // 32, 2, 4, 4).permute(2, 0, 4, 1, 3, 5).reshape(-1, 2048)

    def hm(w):
        h = w.to(torch.bfloat16).float()
        return h, (w - h).to(torch.bfloat16).float()

    def proj_stage_blocks(w, per_stage):                  # [h, m] per row block, zero padded to whole 12-block stages
        h, m = (rowblocks(t) for t in hm(w))
        t = torch.stack([h, m], 1).reshape(-1, 1024)
        pad = (-t.shape[0]) % (2 * per_stage) if per_stage * 2 == 12 else (12 - t.shape[0] % 12) % 12
        return torch.cat([t, torch.zeros(pad, 1024, device=dev)], 0)

    npair = d_ffn // 32
    w1h = rowblocks(w1.to(torch.bfloat16).float()).reshape(npair, 2048)            # W1(q0) | W1(q1) of a pair
    w2h = w2pairs(w2.to(torch.bfloat16).float())                                    # (npair, 2048)
    ffn = torch.cat([w1h, w2h], 1)                                                  # (npair, 4 blocks)
    ffn = torch.cat([ffn, torch.zeros((-npair) % 3, 4096, device=dev)], 0).reshape(-1)
    blocks = [proj_stage_blocks(wo, 4).reshape(-1), ffn]
    if wv is not None:
        blocks += [proj_stage_blocks(wv, 4).reshape(-1), proj_stage_blocks(wp, 6).reshape(-1)]
    out = torch.cat(blocks, 0).to(torch.bfloat16).contiguous().view(torch.int16).reshape(-1)
    need = int(lib().msm_encoder_block_lp_stream_bytes(d_ffn, 0 if wp is None else wp.shape[0]))
    if out.numel() * 2 != need:
        raise RuntimeError(f"pack_encoder_block_lp: built {out.numel() * 2} bytes, the kernel expects {need}")
    return out


def encoder_block_lp(attn, src, wstream, small, d_ffn, proj_width, *, pos=None, tokens_per_image=None, want_next=True,
                     value_heads=0, eps=1e-5):
    """encoder_block in the low-precision mode on the K = 32 kernel (wstream from pack_encoder_block_lp); fp32 in, fp32 out."""
    _c(attn, "attn"), _c(src, "src"), _c(wstream, "wstream", torch.int16), _c(small, "small"), _c(pos, "pos")
    B, S, C = src.shape
    src_out = torch.empty_like(src)
    value_out = proj_out = None
    if want_next:
        value_out = torch.empty((B, value_heads, S, C // value_heads), device=src.device, dtype=torch.float32) \
            if value_heads else torch.empty_like(src)
        proj_out = torch.empty((B, S, proj_width), device=src.device, dtype=torch.float32)
    rc = lib().msm_encoder_block_lp_fwd(_p(attn), _p(src), _p(wstream), _p(small), _p(pos), _p(src_out), _p(value_out), _p(proj_out),
                                        B * S, tokens_per_image or S, d_ffn, proj_width, int(value_heads), eps, _stream())
    check(rc, "msm_encoder_block_lp_fwd")
    return src_out, value_out, proj_out


# ---- the bf16 plan's encoder layers with head-major bf16 activations (csrc/enc_lp.hip) ------------------------------------
def _korder_L(K, device):
    """k order "L" of a K-wide contraction whose B operand comes from layout-L registers (lane (token, lq) holds features
    fb*16 + lq*4 + r): 32-wide group G, lane quarter kq, element j  <->  column (2G + (j >> 2))*16 + 4 kq + (j & 3)."""
    G = torch.arange(K // 32, device=device).view(-1, 1, 1)
    kq = torch.arange(4, device=device).view(1, -1, 1)
    j = torch.arange(8, device=device).view(1, 1, -1)
    return (2 * G + (j >> 2)) * 16 + 4 * kq + (j & 3)


def _korder_natural(K, device):
    G = torch.arange(K // 32, device=device).view(-1, 1, 1)
    kq = torch.arange(4, device=device).view(1, -1, 1)
    j = torch.arange(8, device=device).view(1, 1, -1)
    return 32 * G + 8 * kq + j


def _frag_blocks(w, korder):
    """w (R, K) -> (R/16, K/32, 512): 1-KiB A-operand blocks of v_mfma_f32_16x16x32_bf16, block[rb][G][kq*16 + i][j] =
    w[rb*16 + i][korder[G][kq][j]]."""
    R, K = w.shape
    t = w.reshape(R // 16, 16, K)[:, :, korder]              # (rb, i, G, kq, j)
    return t.permute(0, 2, 3, 1, 4).reshape(R // 16, K // 32, 512)


def _hl(w):
    h = w.to(torch.bfloat16).float()
    return h, (w - h).to(torch.bfloat16).float()


def _value_row_perm(device):
    """Row (16 rb + 4 lq + r) of the packed value_proj = value feature head*8 + dim with head = 4 (rb >> 1) + lq,
    dim = 4 (rb & 1) + r: a lane's row blocks 2j, 2j + 1 are the eight dims of one head (one 16-byte store)."""
    rb = torch.arange(4, device=device).view(-1, 1, 1)
    lq = torch.arange(4, device=device).view(1, -1, 1)
    r = torch.arange(4, device=device).view(1, 1, -1)
    return ((4 * (rb >> 1) + lq) * 8 + 4 * (rb & 1) + r).reshape(-1)


def _proj_row_perm(heads, LP, device):
    """Packed row order of the [sampling_offsets | attention_weights] projection of the bf16 plan: the offsets of all heads (row
    24 head + c), then the logits (192 + 12 head + c) -- the reference's own row order (ms_deform_attn.py:47-48, 99-101), so a
    16-row block of the MFMA output is all offsets or all logits (csrc/enc_lp.hip, store_proj_rb).  (Round 4 interleaved them per head.)"""
    return torch.arange(heads * 3 * LP, device=device)


def _proj_row_perm_per_head(heads, LP, device):
    """Row m*36 + c of the per-head projection blocks of msm_msdeform_attn_enc_lp_fused_fwd = reference row m*2LP + c (offsets,
    c < 2LP) or heads*2LP + m*LP + c - 2LP (logits)."""
    m = torch.arange(heads, device=device).view(-1, 1)
    c = torch.arange(3 * LP, device=device).view(1, -1)
    return torch.where(c < 2 * LP, m * 2 * LP + c, heads * 2 * LP + m * LP + c - 2 * LP).reshape(-1)


PROJ_REC_FLOATS = 30     # the bf16 plan's sampling projection: 120 bytes per (image, head, token) = 24 fp32 offsets + 12 fp16 logits, plane-major
                         # per (image, head) (csrc/enc_lp.hip, EH_REC); tensors are typed (B, 8, S, 30) float32 for their size only


def pack_encoder_block_hm(wo, w1, w2, wv=None, wp=None, ffn_f16=False):
    """One encoder layer's matrices as the weight stream of msm_encoder_block_hm_fwd (include/msm_hip.h): resident block
    [output_proj | next layer's value_proj] as [h, l] bf16 pairs, linear1 / linear2 as single bf16 copies -- ``ffn_f16``: as IEEE
    halves (precision "f16") --, four pairs of 16-wide hidden blocks per 32-KiB stage, then (wv / wp given) the next layer's
    sampling projection in (head, 36) row order as [h, l] pairs, eight row blocks per stage.  Returns an int16 tensor (bit patterns)."""
    dev = wo.device
    d_ffn = w1.shape[0]
    if wo.shape != (64, 64) or w1.shape[1] != 64 or tuple(w2.shape) != (64, d_ffn) or d_ffn % 32:
        raise RuntimeError("pack_encoder_block_hm: d_model 64, d_ffn a multiple of 32")
    if (wv is None) != (wp is None) or (wp is not None and tuple(wp.shape) != (288, 64)):
        raise RuntimeError("pack_encoder_block_hm: wv and wp (288, 64) go together")
    pad = (-d_ffn) % 128

    def pair_hl(w, korder):                                   # (R, 64) -> (R/16, 2, 2, 512): [rb][G][h, l]
        h, l = _hl(w)
        return torch.stack([_frag_blocks(h, korder), _frag_blocks(l, korder)], 2)

    kL, kn = _korder_L(64, dev), _korder_natural(64, dev)
    res = [pair_hl(wo, kn).reshape(-1)]
    res.append(pair_hl(wv[_value_row_perm(dev)], kL).reshape(-1) if wv is not None else torch.zeros(16 * 512, device=dev))
    w1p = torch.cat([w1, torch.zeros(pad, 64, device=dev)], 0)
    w2p = torch.cat([w2, torch.zeros(64, pad, device=dev)], 1)
    npair = (d_ffn + pad) // 32
    b1 = _frag_blocks(w1p, kL).reshape(npair, 4 * 512)                        # [P][q][G][512]
    b2 = _frag_blocks(w2p, _korder_L(d_ffn + pad, dev)).permute(1, 0, 2).reshape(npair, 4 * 512)     # [P][ob][512]
    bits = lambda t, dt: t.to(dt).contiguous().view(torch.int16)             # fp32 -> 16-bit patterns (round to nearest even)
    parts = [bits(torch.cat(res), torch.bfloat16), bits(torch.cat([b1, b2], 1).reshape(-1), torch.float16 if ffn_f16 else torch.bfloat16)]
    if wp is not None:
        pj = pair_hl(wp[_proj_row_perm(8, 12, dev)], kL).reshape(-1)         # 18 row blocks x 4 KiB
        parts.append(bits(torch.cat([pj, torch.zeros(3 * 16384 - pj.numel(), device=dev)]), torch.bfloat16))
    out = torch.cat(parts).contiguous()
    assert out.numel() * 2 == lib().msm_encoder_block_hm_stream_bytes(d_ffn, int(wp is not None))
    return out


def pack_encoder_prologue_hm(wv, wp, bv, bp):
    """Layer 0's value_proj (64, 64) and [sampling_offsets | attention_weights] (288, 64) as the weight blocks and bias vector of
    msm_encoder_prologue_hm_fwd: the blocks of pack_encoder_block_hm, value first.  Returns (int16 blocks, float32 small)."""
    dev = wv.device
    if tuple(wv.shape) != (64, 64) or tuple(wp.shape) != (288, 64):
        raise RuntimeError("pack_encoder_prologue_hm: value_proj (64, 64) and a (288, 64) sampling projection")
    kL = _korder_L(64, dev)

    def pair_hl(w):                                           # (R, 64) -> [rb][G][h, l][512]
        h, l = _hl(w)
        return torch.stack([_frag_blocks(h, kL), _frag_blocks(l, kL)], 2).reshape(-1)

    blocks = torch.cat([pair_hl(wv[_value_row_perm(dev)]), pair_hl(wp[_proj_row_perm(8, 12, dev)])]).to(torch.bfloat16).contiguous().view(torch.int16)
    assert blocks.numel() * 2 == lib().msm_encoder_prologue_hm_weight_bytes()
    small = torch.cat([bv[_value_row_perm(dev)], bp[_proj_row_perm(8, 12, dev)]]).contiguous()
    return blocks, small


def encoder_prologue_hm(raw, stats, gn_params, level_starts, blocks, small, pos, *, groups=32, eps=1e-5):
    """encoder_prologue for the bf16 plan with its projections on the bf16 matrix pipe (msm_encoder_prologue_hm_fwd): normalises raw
    IN PLACE (-> src), returns (src, value (B,8,S,8) fp16, proj (B,8,S,36) fp16).  blocks / small: pack_encoder_prologue_hm."""
    _c(raw, "raw"), _c(stats, "stats", torch.float64), _c(gn_params, "gn_params"), _c(blocks, "blocks", torch.int16), _c(small, "small"), _c(pos, "pos")
    B, S, C = raw.shape
    L = len(level_starts) - 1
    if C != 64 or tuple(stats.shape) != (L, B, 64, 2) or tuple(gn_params.shape) != (L, 2, 64) or tuple(pos.shape) != (S, 64) or small.numel() != 352:
        raise RuntimeError("encoder_prologue_hm: inconsistent shapes")
    value = torch.empty((B, 8, S, 8), device=raw.device, dtype=torch.float16)
    proj = torch.empty((B, 8, S, PROJ_REC_FLOATS), device=raw.device, dtype=torch.float32)
    ls = (ctypes.c_int32 * (L + 1))(*[int(v) for v in level_starts])
    rc = lib().msm_encoder_prologue_hm_fwd(_p(raw), _p(stats), _p(gn_params), ctypes.cast(ls, ctypes.c_void_p), L, int(groups), float(eps),
                                           _p(blocks), _p(small), _p(pos), _p(raw), _p(value), _p(proj), B, S, _stream())
    check(rc, "msm_encoder_prologue_hm_fwd")
    return raw, value, proj


def pack_encoder_block_hm_small(bo, g1, be1, b1, b2, g2, be2, bv=None, bp=None):
    """The fp32 parameter vector of msm_encoder_block_hm_fwd (value_proj / projection biases in the packed row orders, linear1
    bias zero padded to whole stages)."""
    dev = bo.device
    d_ffn = b1.numel()
    bvp = bv[_value_row_perm(dev)] if bv is not None else torch.zeros(64, device=dev)
    bpp = bp[_proj_row_perm(8, 12, dev)] if bp is not None else torch.zeros(288, device=dev)
    out = torch.cat([bo, g1, be1, b2, g2, be2, bvp, bpp, b1, torch.zeros((-d_ffn) % 128, device=dev)]).contiguous()
    assert out.numel() == lib().msm_encoder_block_hm_small_floats(d_ffn)
    return out


def pack_msda_proj_lp(wp, bp, heads=8, n_levels=3, n_points=4):
    """[sampling_offsets ; attention_weights] weight (heads*L*P*3, 64) and bias -> the per-head [h, l] bf16 fragment stream
    (int16, 12 KiB per head) and bias table (heads, 48) of msm_msdeform_attn_enc_lp_fused_fwd."""
    LP = n_levels * n_points
    if tuple(wp.shape) != (heads * LP * 3, 64) or bp.numel() != wp.shape[0] or LP != 12:
        raise RuntimeError("pack_msda_proj_lp: the shipped geometry only (3 levels x 4 points)")
    dev = wp.device
    perm = _proj_row_perm_per_head(heads, LP, dev)
    rows = torch.zeros(heads, 48, 64, device=dev)
    bias = torch.zeros(heads, 48, device=dev)
    rows[:, :3 * LP] = wp[perm].reshape(heads, 3 * LP, 64)
    bias[:, :3 * LP] = bp[perm].reshape(heads, 3 * LP)
    kL = _korder_L(64, dev)
    h, l = _hl(rows.reshape(heads * 48, 64))
    blocks = torch.stack([_frag_blocks(h, kL), _frag_blocks(l, kL)], 2)       # (heads*3, 2, 2, 512)
    return blocks.reshape(-1).to(torch.bfloat16).contiguous().view(torch.int16), bias.contiguous()


def proj_to_head_major_records(proj, heads=8, LP=12):
    """(B, S, heads*LP*3) fp32 in the reference's [offsets | logits] column order -> the bf16 plan's sampling projection
    (B, heads, S, 30) float32-TYPED (120 bytes per token; NOT a (.., S, 30) array): per (image, head) six planes [S][4 floats] of fp32
    offsets then three planes [S][4 halves] of fp16 logits (csrc/enc_lp.hip, EH_REC).  Torch ops: tests and the unfused front end
    only; the fused prologue writes this layout itself."""
    B, S, W = proj.shape
    off = proj[..., :heads * 2 * LP].reshape(B, S, heads, 2 * LP // 4, 4).permute(0, 2, 3, 1, 4).reshape(B, heads, -1)          # (B, heads, 6 S 4)
    lg = proj[..., heads * 2 * LP:].reshape(B, S, heads, LP // 4, 4).permute(0, 2, 3, 1, 4).to(torch.float16).reshape(B, heads, -1)
    return torch.cat([off, lg.contiguous().view(torch.float32)], -1).view(B, heads, S, PROJ_REC_FLOATS).contiguous()


def proj_records_to_columns(rec, heads=8, LP=12):
    """Inverse of proj_to_head_major_records (the logits come back as the fp16 values the planes hold): (B, S, heads*LP*3) fp32."""
    B, M, S, _ = rec.shape
    flat = rec.reshape(B, M, S * PROJ_REC_FLOATS)
    off = flat[..., :S * 2 * LP].reshape(B, M, 2 * LP // 4, S, 4).permute(0, 3, 1, 2, 4).reshape(B, S, M * 2 * LP)
    lg = flat[..., S * 2 * LP:].contiguous().view(torch.float16).reshape(B, M, LP // 4, S, 4).permute(0, 3, 1, 2, 4).reshape(B, S, M * LP).float()
    return torch.cat([off, lg], -1).contiguous()


_GLUE_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def bias_act_nhwc_(x, bias, residual=None, relu=True):
    """In place on a channels_last (B, C, H, W) map (fp32 or bfloat16): x = act(x + bias[c] (+ residual)) in one pass
    (msm_bias_act_nhwc) -- the bias kernel MIOpen appends to a convolution, F.relu, the residual add and its ReLU of a ResNet block
    as ONE launch.  bias (C,) and residual (same shape and memory format) in x's dtype.  Returns x."""
    if x.dtype not in _GLUE_DTYPES or not x.is_cuda:
        raise RuntimeError("bias_act_nhwc_: a float32, bfloat16 or float16 map on the GPU")
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("bias_act_nhwc_: x must be (B, C, H, W) in channels_last memory")
    B, C, H, W = x.shape
    if bias.dtype != x.dtype or tuple(bias.shape) != (C,) or not bias.is_contiguous() or bias.device != x.device:
        raise RuntimeError("bias_act_nhwc_: bias must be (C,) in x's dtype on x's device")
    if residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape or residual.device != x.device
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("bias_act_nhwc_: residual must match x (shape, dtype, channels_last)")
    check(lib().msm_bias_act_nhwc(_p(x), _p(bias), _p(residual), 1 if relu else 0, B * H * W, C, _GLUE_DTYPES[x.dtype], _stream()),
          "msm_bias_act_nhwc")
    return x


def nhwc_to_nchw_f32(x):
    """A channels_last (B, C, H, W) map (fp32 or bfloat16) as contiguous NCHW fp32 planes in one pass (msm_nhwc_to_nchw_f32)."""
    if x.dtype not in _GLUE_DTYPES or not x.is_cuda or x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("nhwc_to_nchw_f32: a channels_last float32 / bfloat16 / float16 (B, C, H, W) map on the GPU")
    B, C, H, W = x.shape
    out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    check(lib().msm_nhwc_to_nchw_f32(_p(x), _p(out), B, C, H * W, _GLUE_DTYPES[x.dtype], _stream()), "msm_nhwc_to_nchw_f32")
    return out


def ucn_embedding_tail(a, b=None, size=None, norms=1, eps=1e-12):
    """The tail of the UCN RGB-D backbone in one pass (msm_ucn_embedding_tail): a, b (B, 64, h, w) fp32 channels_last maps of the two towers
    (b None: one tower) -> (B, 64, H, W) contiguous fp32 = N(..N(upsample_bilinear(a) + upsample_bilinear(b))), align_corners=True,
    N = F.normalize over the channels applied ``norms`` (0..2) times."""
    if a.dtype != torch.float32 or not a.is_cuda or a.dim() != 4 or a.shape[1] != 64:
        raise RuntimeError("ucn_embedding_tail: (B, 64, h, w) float32 maps on the GPU")
    a = a if a.is_contiguous(memory_format=torch.channels_last) else a.contiguous(memory_format=torch.channels_last)
    if b is not None:
        if b.shape != a.shape or b.dtype != a.dtype or b.device != a.device:
            raise RuntimeError("ucn_embedding_tail: the two towers' maps must match")
        b = b if b.is_contiguous(memory_format=torch.channels_last) else b.contiguous(memory_format=torch.channels_last)
    B, _, h, w = a.shape
    H, W = int(size[0]), int(size[1])
    out = torch.empty((B, 64, H, W), device=a.device, dtype=torch.float32)
    check(lib().msm_ucn_embedding_tail(_p(a), _p(b), _p(out), B, h, w, H, W, int(norms), float(eps), _stream()), "msm_ucn_embedding_tail")
    return out


def to_f16(t):
    """fp32 -> fp16 (round to nearest even, clamped to the half range) on the HIP path (msm_f32_to_f16)."""
    _c(t, "t")
    out = torch.empty(t.shape, device=t.device, dtype=torch.float16)
    check(lib().msm_f32_to_f16(_p(t), _p(out), t.numel(), _stream()), "msm_f32_to_f16")
    return out


def encoder_block_hm(attn_hm, src, wstream, small, d_ffn, *, pos=None, want_next=True, eps=1e-5, ffn_f16=False):
    """One encoder-layer tail of the bf16 plan: attn_hm (B, 8, S, 8) fp16, src (B, S, 64) fp32 -> (src_out fp32, and for the
    NEXT layer value_hm (B, 8, S, 8) fp16 and the sampling records proj_hm (B, 8, S, 30) float32-typed (24 fp32 offsets + 12 fp16
    logits), or None, None)."""
    _c(attn_hm, "attn_hm", torch.float16), _c(src, "src"), _c(wstream, "wstream", torch.int16), _c(small, "small"), _c(pos, "pos")
    B, S, C = src.shape
    if C != 64 or tuple(attn_hm.shape) != (B, 8, S, 8):
        raise RuntimeError("encoder_block_hm: src (B, S, 64) and attn_hm (B, 8, S, 8)")
    if want_next and (pos is None or tuple(pos.shape) != (S, 64)):
        raise RuntimeError("encoder_block_hm: the next layer's projection needs pos (S, 64)")
    if wstream.numel() * 2 != lib().msm_encoder_block_hm_stream_bytes(int(d_ffn), int(want_next)):
        raise RuntimeError("encoder_block_hm: wstream does not match d_ffn / want_next (pack_encoder_block_hm)")
    src_out = torch.empty_like(src)
    value_out = torch.empty_like(attn_hm) if want_next else None
    proj_out = torch.empty((B, 8, S, PROJ_REC_FLOATS), device=src.device, dtype=torch.float32) if want_next else None
    rc = lib().msm_encoder_block_hm_fwd(_p(attn_hm), _p(src), _p(wstream), _p(small), _p(pos if want_next else None), _p(src_out), _p(value_out),
                                        _p(proj_out), B * S, S, int(d_ffn), float(eps), int(bool(ffn_f16)), _stream())
    check(rc, "msm_encoder_block_hm_fwd")
    return src_out, value_out, proj_out


def ms_deform_attn_encoder_lp(value_hm, spatial_shapes, level_start_index, proj_hm, n_points=4):
    """Encoder self-attention gather of the bf16 plan: value_hm (B, 8, S, 8) fp16, proj_hm (B, 8, S, 30) 120-byte records (fp32 offsets,
    logits).  Returns attn_hm (B, 8, S, 8) fp16."""
    _c(value_hm, "value_hm", torch.float16), _c(proj_hm, "proj_hm", torch.float32)
    _c(spatial_shapes, "spatial_shapes", torch.int64), _c(level_start_index, "level_start_index", torch.int64)
    B, M, S, D = value_hm.shape
    if tuple(proj_hm.shape) != (B, M, S, PROJ_REC_FLOATS):
        raise RuntimeError("ms_deform_attn_encoder_lp: proj_hm must be (B, heads, S, 30) float32-typed 120-byte records")
    out = torch.empty_like(value_hm)
    rc = lib().msm_msdeform_attn_enc_lp_fwd(_p(value_hm), _p(spatial_shapes), _p(level_start_index), _p(proj_hm), _p(out), B, S, M, D,
                                            spatial_shapes.shape[0], int(n_points), _stream())
    check(rc, "msm_msdeform_attn_enc_lp_fwd")
    return out


def ms_deform_attn_encoder_lp_fused(value_hm, spatial_shapes, level_start_index, src, pos, wpack, bpack, n_points=4):
    """The same gather with the sampling projection of src + pos computed in the kernel (wpack / bpack from pack_msda_proj_lp)."""
    _c(value_hm, "value_hm", torch.float16), _c(src, "src"), _c(pos, "pos"), _c(wpack, "wpack", torch.int16), _c(bpack, "bpack")
    _c(spatial_shapes, "spatial_shapes", torch.int64), _c(level_start_index, "level_start_index", torch.int64)
    B, M, S, D = value_hm.shape
    if tuple(src.shape) != (B, S, M * D) or tuple(pos.shape) != (S, M * D):
        raise RuntimeError("ms_deform_attn_encoder_lp_fused: src (B, S, 64), pos (S, 64)")
    out = torch.empty_like(value_hm)
    rc = lib().msm_msdeform_attn_enc_lp_fused_fwd(_p(value_hm), _p(spatial_shapes), _p(level_start_index), _p(src), _p(pos), _p(wpack),
                                                  _p(bpack), _p(out), B, S, M, D, spatial_shapes.shape[0], int(n_points), _stream())
    check(rc, "msm_msdeform_attn_enc_lp_fused_fwd")
    return out


def encoder_block_split(attn, src, wstream, small, d_ffn, proj_width, *, pos=None, tokens_per_image=None, want_next=True,
                        value_heads=0, eps=1e-5):
    """encoder_block in fp32 accuracy on the bf16 matrix pipe (wstream from pack_encoder_block_split): fp32 in, fp32 out."""
    _c(attn, "attn"), _c(src, "src"), _c(wstream, "wstream", torch.int16), _c(small, "small"), _c(pos, "pos")
    B, S, C = src.shape
    src_out = torch.empty_like(src)
    value_out = proj_out = None
    if want_next:
        value_out = torch.empty((B, value_heads, S, C // value_heads), device=src.device, dtype=torch.float32) \
            if value_heads else torch.empty_like(src)
        proj_out = torch.empty((B, S, proj_width), device=src.device, dtype=torch.float32)
    rc = lib().msm_encoder_block_split_fwd(_p(attn), _p(src), _p(wstream), _p(small), _p(pos), _p(src_out), _p(value_out), _p(proj_out),
                                           B * S, tokens_per_image or S, d_ffn, proj_width, int(value_heads), eps, _stream())
    check(rc, "msm_encoder_block_split_fwd")
    return src_out, value_out, proj_out


def encoder_block(attn, src, wstream, small, d_ffn, proj_width, *, pos=None, tokens_per_image=None, want_next=True,
                  value_heads=0, eps=1e-5):
    """One fused encoder-layer tail.  attn/src (B,S,64).  Returns (src_out, value_out, proj_out) with the
    last two None when want_next is False.  value_heads = h > 0: value_out is head-major (B,h,S,64/h)."""
    _c(attn, "attn"), _c(src, "src"), _c(wstream, "wstream"), _c(small, "small"), _c(pos, "pos")
    B, S, C = src.shape
    M = B * S
    src_out = torch.empty_like(src)
    value_out = proj_out = None
    if want_next:
        value_out = torch.empty((B, value_heads, S, C // value_heads), device=src.device, dtype=torch.float32) \
            if value_heads else torch.empty_like(src)
        # proj_width == 0: the next layer's gather computes its own sampling projection (ms_deform_attn_encoder_fused)
        proj_out = torch.empty((B, S, proj_width), device=src.device, dtype=torch.float32) if proj_width else None
    rc = lib().msm_encoder_block_fwd(_p(attn), _p(src), _p(wstream), _p(small), _p(pos), _p(src_out), _p(value_out),
                                     _p(proj_out), M, tokens_per_image or S, d_ffn, proj_width, int(value_heads), eps, _stream())
    check(rc, "msm_encoder_block_fwd")
    return src_out, value_out, proj_out
