"""Kernel-level parity: every C-ABI entry point against the CPU oracle / plain torch fp32 on the
same seeded inputs.  Needs a real MI355X (pytest -m gpu)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import msm_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from unseenobjectswithmeanshift_amd import ops as _ops
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, ref, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(got.cpu(), ref, rtol=rtol, atol=atol)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(800, 256, 256), (37, 3, 256), (800, 2048, 256), (50400 // 8, 288, 64),
                                   (100, 512, 256), (130, 70, 36)])
def test_gemm_linear(M, N, K):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = F.linear(a, w, b)
    close(ops().gemm(a.to(DEV), w.to(DEV), b.to(DEV)), ref)
    close(ops().gemm(a.to(DEV), w.to(DEV), b.to(DEV), act="relu"), F.relu(ref))
    close(ops().gemm(a.to(DEV), w.to(DEV)), F.linear(a, w))


def test_gemm_a2_broadcast_and_splitk():
    B, L, K, N = 3, 100, 256, 256
    a, a2, w, b = rnd(B, L, K, seed=1), rnd(L, K, seed=2), rnd(N, K, seed=3, scale=K ** -0.5), rnd(N, seed=4)
    close(ops().gemm(a.to(DEV), w.to(DEV), b.to(DEV), a2=a2.to(DEV)), F.linear(a + a2, w, b))
    a2f = rnd(B, L, K, seed=5)
    close(ops().gemm(a.to(DEV), w.to(DEV), b.to(DEV), a2=a2f.to(DEV)), F.linear(a + a2f, w, b))
    # split-K raw parts + layernorm consumer
    K2 = 2048
    h, w2, b2 = rnd(B, L, K2, seed=6), rnd(N, K2, seed=7, scale=K2 ** -0.5), rnd(N, seed=8)
    parts = ops().gemm(h.to(DEV), w2.to(DEV), split_k=8)
    assert parts.shape == (8, B, L, N)
    close(parts.sum(0), F.linear(h, w2), rtol=1e-4, atol=1e-4)
    x, g1, be1, g2, be2 = rnd(B, L, N, seed=9), 1 + 0.1 * rnd(N, seed=10), rnd(N, seed=11), 1 + 0.1 * rnd(N, seed=12), rnd(N, seed=13)
    y, y2 = ops().layernorm(x.to(DEV), g1.to(DEV), be1.to(DEV), parts=parts, bias=b2.to(DEV), l2norm=True,
                            g2=g2.to(DEV), b2=be2.to(DEV))
    r = F.layer_norm(x + F.linear(h, w2, b2), (N,), g1, be1)
    r = r / r.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    close(y, r, atol=2e-5)
    close(y2, F.layer_norm(r, (N,), g2, be2), atol=1e-4)


@pytest.mark.parametrize("E", [64, 256])
def test_layernorm_plain(E):
    x, t, g, b = rnd(1000, E, seed=1), rnd(1, 1000, E, seed=2), 1 + 0.1 * rnd(E, seed=3), rnd(E, seed=4)
    y = ops().layernorm(x.to(DEV), g.to(DEV), b.to(DEV), parts=t.to(DEV))
    close(y, F.layer_norm(x + t[0], (E,), g, b), atol=2e-5)


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(2, 2048, 15, 20, 64), (2, 256, 16, 24, 64), (2, 512, 2, 3, 64), (1, 64, 60, 80, 256)])
def test_conv1x1_layouts(B, Cin, H, W, Cout):
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, seed=2, scale=Cin ** -0.5), rnd(Cout, seed=3)
    ref = F.conv2d(x, w[:, :, None, None], b)
    tok = ops().conv1x1_nchw_to_tokens(x.to(DEV), w.to(DEV), b.to(DEV))
    close(tok, ref.flatten(2).transpose(1, 2), atol=1e-4)
    back = ops().conv1x1_tokens_to_nchw(tok, torch.eye(Cout, device=DEV).contiguous(), None)
    close(back, ref.flatten(2), atol=1e-4)
    close(ops().transpose_last2(tok), ref.flatten(2), atol=1e-4)


def test_conv3x3_groupnorm_fpn_path():
    B, C, H, W = 2, 64, 16, 24
    x, w = rnd(B, C, H, W, seed=1), rnd(C, C, 3, 3, seed=2, scale=(9 * C) ** -0.5)
    g, be = 1 + 0.1 * rnd(C, seed=3), rnd(C, seed=4)
    up = rnd(B, C, H // 2, W // 2, seed=5)
    tok = x.flatten(2).transpose(1, 2).contiguous().to(DEV)
    uptok = up.flatten(2).transpose(1, 2).contiguous().to(DEV)
    # lateral: GN(x) + bilinear up
    y = ops().groupnorm_tokens(tok, g.to(DEV), be.to(DEV), H, W, up=uptok, up_hw=(H // 2, W // 2))
    ref = F.group_norm(x, 32, g, be) + F.interpolate(up, size=(H, W), mode="bilinear", align_corners=False)
    close(y, ref.flatten(2).transpose(1, 2), atol=2e-5)
    # output conv: 3x3 + GN + relu
    wt = w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().to(DEV)
    c = ops().conv3x3_tokens(tok, wt, H, W)
    refc = F.conv2d(x, w, None, padding=1)
    close(c, refc.flatten(2).transpose(1, 2), atol=1e-4)
    y = ops().groupnorm_tokens(c, g.to(DEV), be.to(DEV), H, W, relu=True)
    close(y, F.relu(F.group_norm(refc, 32, g, be)).flatten(2).transpose(1, 2), atol=1e-4)


@pytest.mark.parametrize("npf,H,W", [(128, 15, 20), (32, 30, 40), (32, 3, 2)])
def test_pos_embed(npf, H, W):
    ref = O.position_embedding_sine(1, H, W, npf)[0]
    close(ops().pos_embed_sine(H, W, npf, DEV), ref, atol=2e-6)
    add = rnd(2 * npf, seed=1)
    tok = ops().pos_embed_sine(H, W, npf, DEV, layout="tokens", add_c=add.to(DEV))
    close(tok, ref.flatten(1).t() + add, atol=2e-6)


# ---------------------------------------------------------------------------------------------
def ref_mask_step(e, f, tgt):
    mask = torch.einsum("bqc,bchw->bqhw", e, f)
    m = F.interpolate(mask, size=tgt, mode="bilinear", align_corners=False)
    return mask, (m.sigmoid().flatten(2) < 0.5)


@pytest.mark.parametrize("nc", ["2", "1"])
@pytest.mark.parametrize("B,Q,H,W,pool", [(2, 100, 16, 24, 2), (2, 100, 16, 24, 4), (2, 100, 16, 24, 8),
                                          (1, 100, 120, 160, 8), (1, 100, 120, 160, 4), (1, 100, 120, 160, 2),
                                          (1, 300, 48, 64, 4), (2, 20, 8, 8, 2), (2, 100, 16, 24, 1), (1, 100, 60, 80, 1)])
def test_mask_logits(B, Q, H, W, pool, nc, lib_option):
    lib_option("MASK_NC", int(nc))          # wave tile 2x32 (8-byte loads) or 2x16 (4-byte loads)
    C = 256
    e, f = rnd(B, Q, C, seed=1, scale=0.3), rnd(B, C, H, W, seed=2)
    tgt = (H // pool, W // pool)
    mask_ref, attn_ref = ref_mask_step(e, f, tgt)
    for want_mask, sparse in ((True, False), (False, False), (False, True)):
        mask, attn, row_any = ops().mask_logits(e.to(DEV), f.to(DEV), want_mask=want_mask, target_size=tgt, sparse=sparse)
        if want_mask:
            close(mask, mask_ref, rtol=1e-4, atol=1e-4)
        # bits may differ only where the pooled logit is within rounding of zero
        got = attn.cpu().bool()
        diff = got != attn_ref
        if diff.any():
            m = F.interpolate(mask_ref, size=tgt, mode="bilinear", align_corners=False).flatten(2)
            assert m[diff].abs().max() < 1e-4
        assert diff.float().mean() < 1e-4
        assert torch.equal(row_any.cpu().bool(), ~got.all(-1))
    mask, attn, row_any = ops().mask_logits(e.to(DEV), f.to(DEV), want_mask=True, target_size=None)
    close(mask, mask_ref, rtol=1e-4, atol=1e-4)
    assert attn is None and row_any is None


@pytest.mark.parametrize("kernel", ["default", "plain"])
@pytest.mark.parametrize("B,Q,H,W,pool", [(2, 100, 16, 24, 2), (2, 100, 16, 24, 4), (2, 100, 16, 24, 8), (8, 100, 120, 160, 8),
                                          (1, 100, 120, 160, 4), (2, 100, 120, 160, 2), (1, 300, 48, 64, 4), (2, 20, 8, 8, 2),
                                          (2, 100, 16, 24, 1), (1, 100, 60, 80, 1), (3, 100, 30, 40, 2), (1, 37, 18, 22, 2)])
def test_mask_logits_folded_form(B, Q, H, W, pool, kernel, lib_option):
    """The folded form of the step (modeling.FoldedMaskFeatures): 64 channels, the embedding is the leading 64 columns of a
    256-wide buffer, a per-query bias starts every logit.  "default": the library's choice (with 100 queries the last four on
    the 4x4x1 MFMA); "plain": the fallback kernel without that block (MSM_OPT_MASK_KERNEL = 5)."""
    if kernel == "plain":
        lib_option("MASK_KERNEL", 5)
    C = 64
    wide = rnd(B, Q, 256, seed=1, scale=0.3)
    e, qb = wide[..., :C], wide[..., 64]
    f = rnd(B, C, H, W, seed=2)
    tgt = (H // pool, W // pool)
    full = torch.einsum("bqc,bchw->bqhw", e.double(), f.double()) + qb.double()[..., None, None]
    pooled = F.interpolate(full.float(), size=tgt, mode="bilinear", align_corners=False)
    attn_ref = pooled.sigmoid().flatten(2) < 0.5
    wd, fd = wide.to(DEV), f.to(DEV)
    for want_mask, sparse in ((True, False), (False, False), (False, True)):
        mask, attn, row_any = ops().mask_logits(wd[..., :C], fd, want_mask=want_mask, target_size=tgt, sparse=sparse, qbias=wd[..., 64])
        if want_mask:
            close(mask, full.float(), rtol=1e-4, atol=1e-4)
        got = attn.cpu().bool()
        diff = got != attn_ref
        if diff.any():                                # bits may differ only where the pooled logit is within rounding of zero
            assert pooled.flatten(2)[diff].abs().max() < 1e-4
        assert diff.float().mean() <= 1e-4            # attention-mask bit mismatch <= 1e-4
        assert torch.equal(row_any.cpu().bool(), ~attn.cpu().bool().all(-1))
    mask, attn, row_any = ops().mask_logits(wd[..., :C], fd, want_mask=True, target_size=None, qbias=wd[..., 64])
    close(mask, full.float(), rtol=1e-4, atol=1e-4)
    assert attn is None and row_any is None


@pytest.mark.parametrize("B,Q,H,W", [(2, 100, 16, 24), (8, 100, 120, 160), (1, 37, 48, 64), (3, 100, 56, 56), (2, 112, 16, 40), (1, 300, 48, 64)])
def test_attention_masks_at_key_resolution(B, Q, H, W):
    """msm_pool_mask_taps + msm_attn_mask_pooled: bilinear reduction and channel contraction commute
    (interpolate(einsum(e, F)) = einsum(e, interpolate(F))), so the attention masks of the intermediate predictions come from
    the activation pooled to the key resolutions.  Against the reference order in float64 (einsum at full resolution, then
    F.interpolate, DEC:668-680) and against the full-resolution kernel: bits may differ only where the logit is within
    rounding of zero; the row flags agree with the bits; the pooled activation is F.interpolate of the activation."""
    wide = rnd(B, Q, 256, seed=1, scale=0.3)
    f = rnd(B, 64, H, W, seed=2)
    wd, fd = wide.to(DEV), f.to(DEV)
    sizes = [(H // p, W // p) for p in (8, 4, 2)]
    pooled = ops().pool_mask_taps(fd, sizes)
    # the same launch clears a (B, Q) flag buffer when asked to (pre-filled here through the caching allocator's reuse)
    torch.full((B, Q), 7, device=DEV, dtype=torch.int32)
    pooled2, flags = ops().pool_mask_taps(fd, sizes, zero_rows=Q)
    assert flags.shape == (B, Q) and flags.dtype == torch.int32 and int(flags.abs().sum()) == 0
    assert all(torch.equal(a, b) for a, b in zip(pooled, pooled2))
    full = torch.einsum("bqc,bchw->bqhw", wide[..., :64].double(), f.double()) + wide[..., 64].double()[..., None, None]
    for (th, tw), ap in zip(sizes, pooled):
        assert ap.shape == (B, th * tw, 64)
        ref_p = F.interpolate(f, size=(th, tw), mode="bilinear", align_corners=False).flatten(2).transpose(1, 2)
        close(ap, ref_p, rtol=1e-6, atol=1e-6)
        ref_logit = F.interpolate(full.float(), size=(th, tw), mode="bilinear", align_corners=False).flatten(2)
        attn_ref = ref_logit.sigmoid() < 0.5
        ra0 = torch.zeros(B, Q, device=DEV, dtype=torch.int32)
        for ra in (None, ra0):
            attn, row_any = ops().attn_mask_pooled(wd[..., :64], ap, qbias=wd[..., 64], row_any=ra)
            got = attn.cpu().bool()
            diff = got != attn_ref
            if diff.any():
                assert ref_logit[diff].abs().max() < 1e-4
            assert diff.float().mean() <= 1e-4
            assert torch.equal(row_any.cpu().bool(), ~got.all(-1))
        if (th * tw) % 16 == 0:
            # the bit-packed, blocked form for the fused K/V attention (bits=True) = attn_pack_mask_bits of the byte mask, row flags alike
            # (word 7 of a query's eight is never written: compared on the seven query blocks of a chunk)
            ab, rab = ops().attn_mask_pooled(wd[..., :64], ap, qbias=wd[..., 64], bits=True)
            assert ab.dtype == torch.int16 and ab.shape == (B, (Q + 111) // 112, th * tw // 16, 16, 8)
            want_b = ops().attn_pack_mask_bits(attn)
            nblk = [min(7, (Q - 112 * qc + 15) // 16) for qc in range(ab.shape[1])]
            for qc, nb in enumerate(nblk):
                got_b, exp_b = ab[:, qc, :, :, :nb].cpu(), want_b[:, qc, :, :, :nb].cpu()
                if Q % 16 and qc == len(nblk) - 1:          # rows past Q in the last block: lanes without a query write nothing
                    live = (torch.arange(16)[:, None] + 16 * torch.arange(nb)[None] + 112 * qc) < Q
                    got_b, exp_b = got_b[..., live], exp_b[..., live]
                assert torch.equal(got_b, exp_b)
            assert torch.equal(rab, row_any)
        # f16=True (16-bit plans): embedding and pooled activation as IEEE halves on the 16-bit matrix pipe -- against float64 on the operands
        # as rounded (bits decided wherever the logit is beyond the fp32 accumulation error), close to the fp32 form, both output layouts
        h16 = lambda t: t.to(torch.float16).double()
        ref_h = torch.einsum("bqc,btc->bqt", h16(wide[..., :64]), h16(ap.cpu())) + wide[..., 64].double()[..., None]
        ah, rah = ops().attn_mask_pooled(wd[..., :64], ap, qbias=wd[..., 64], f16=True)
        dh = ah.cpu().bool() != (ref_h < 0)
        if dh.any():
            assert ref_h[dh].abs().max() < 2e-5 * float(ref_h.abs().max())
        assert dh.float().mean() <= 1e-4 and torch.equal(rah.cpu().bool(), ~ah.cpu().bool().all(-1))
        assert float((ah != attn).float().mean()) < 2e-3                          # (the half rounding moves logits by ~1e-3 of their scale)
        if (th * tw) % 16 == 0:
            abh, rabh = ops().attn_mask_pooled(wd[..., :64], ap, qbias=wd[..., 64], bits=True, f16=True)
            want_h = ops().attn_pack_mask_bits(ah)
            for qc, nb in enumerate(nblk):
                got_b, exp_b = abh[:, qc, :, :, :nb].cpu(), want_h[:, qc, :, :, :nb].cpu()
                if Q % 16 and qc == len(nblk) - 1:
                    live = (torch.arange(16)[:, None] + 16 * torch.arange(nb)[None] + 112 * qc) < Q
                    got_b, exp_b = got_b[..., live], exp_b[..., live]
                assert torch.equal(got_b, exp_b)
            assert torch.equal(rabh, rah)
        # f16="x3" (round 6, the 16-bit plans' default): hi + lo IEEE-half operand pairs, three terms per product -- held to the FP32 form's bounds
        # against the float64 reference order (bits may differ only where the logit is within rounding of zero), both output layouts
        a3, ra3 = ops().attn_mask_pooled(wd[..., :64], ap, qbias=wd[..., 64], f16="x3")
        d3 = a3.cpu().bool() != attn_ref
        if d3.any():
            assert ref_logit[d3].abs().max() < 1e-4
        assert d3.float().mean() <= 1e-4 and torch.equal(ra3.cpu().bool(), ~a3.cpu().bool().all(-1))
        assert float((a3 != attn).float().mean()) <= 2e-5                         # (against the fp32 MFMA chain: a few logits within 1e-6 of zero)
        if (th * tw) % 16 == 0:
            ab3, rab3 = ops().attn_mask_pooled(wd[..., :64], ap, qbias=wd[..., 64], bits=True, f16="x3")
            want_3 = ops().attn_pack_mask_bits(a3)
            for qc, nb in enumerate(nblk):
                got_b, exp_b = ab3[:, qc, :, :, :nb].cpu(), want_3[:, qc, :, :, :nb].cpu()
                if Q % 16 and qc == len(nblk) - 1:
                    live = (torch.arange(16)[:, None] + 16 * torch.arange(nb)[None] + 112 * qc) < Q
                    got_b, exp_b = got_b[..., live], exp_b[..., live]
                assert torch.equal(got_b, exp_b)
            assert torch.equal(rab3, ra3)
        _, attn_full, ra_full = ops().mask_logits(wd[..., :64], fd, want_mask=False, target_size=(th, tw), qbias=wd[..., 64])
        d2 = attn_full.cpu() != attn.cpu()
        if d2.any():
            assert ref_logit[d2].abs().max() < 1e-4
        assert d2.float().mean() <= 1e-4
    # a key count that is no multiple of four (7 x 7: the coarsest level of a 224 x 224 crop) takes the bytewise stores
    if H == 56:
        ap = ops().pool_mask_taps(fd, [(7, 7)])[0]
        attn, row_any = ops().attn_mask_pooled(wd[..., :64], ap, qbias=wd[..., 64])
        ref_logit = F.interpolate(full.float(), size=(7, 7), mode="bilinear", align_corners=False).flatten(2)
        diff = attn.cpu().bool() != (ref_logit < 0)
        assert diff.float().mean() <= 1e-4 and torch.equal(row_any.cpu().bool(), ~attn.cpu().bool().all(-1))


def test_mask_logits_tile_choice_is_result_neutral(lib_option):
    e, f = rnd(8, 100, 256, seed=3, scale=0.3).to(DEV), rnd(8, 256, 120, 160, seed=4).to(DEV)
    outs = []
    for nc in ("2", "1"):
        lib_option("MASK_NC", int(nc))
        outs.append(ops().mask_logits(e, f, want_mask=True, target_size=(30, 40)))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Lq,S,masked", [(2, 100, 300, True), (2, 100, 100, False), (1, 100, 4800, True),
                                           (2, 100, 6, True), (1, 300, 1200, True), (2, 20, 37, True)])
def test_hypersphere_attention(B, Lq, S, masked):
    H, E = 8, 256
    q, k, v = rnd(B, Lq, E, seed=1), rnd(B, S, E, seed=2), rnd(B, S, E, seed=3)
    m = None
    row_any = None
    add = None
    if masked:
        g = torch.Generator().manual_seed(4)
        m = torch.rand(B, Lq, S, generator=g) < 0.6
        m[0, 1] = True                      # an all-masked row: must attend everywhere (DEC:618)
        row_any = (~m.all(-1)).to(torch.int32)
        eff = m.clone()
        eff[m.all(-1)] = False
        add = torch.zeros(B, 1, Lq, S)
        add[eff[:, None]] = float("-inf")
        add = add.expand(B, H, Lq, S).reshape(B * H, Lq, S)
    qh = q.view(B, Lq, H, 32).permute(0, 2, 1, 3).reshape(B * H, Lq, 32)
    kh = k.view(B, S, H, 32).permute(0, 2, 1, 3).reshape(B * H, S, 32)
    vh = v.view(B, S, H, 32).permute(0, 2, 1, 3).reshape(B * H, S, 32)
    o, _ = O.hypersphere_attention(qh, kh, vh, add)
    ref = o.view(B, H, Lq, 32).permute(0, 2, 1, 3).reshape(B, Lq, E)
    args = (q.to(DEV), k.to(DEV), v.to(DEV), H)
    kw = dict(masked=None if m is None else m.to(torch.uint8).to(DEV), row_any=None if row_any is None else row_any.to(DEV))
    got = ops().hypersphere_attention(*args, **kw)
    close(got, ref, rtol=1e-4, atol=2e-5)
    # the fallback kernel the host can be told to take: split-K + combine at every length
    from unseenobjectswithmeanshift_amd._lib import option
    with option("ATTN_KERNEL", 3):
        alt = ops().hypersphere_attention(*args, **kw)
    close(alt, ref, rtol=1e-4, atol=2e-5)
    # every workgroup shape of the key-split kernel (two / one / four query blocks per workgroup; four is what batches of >= 48 images take)
    if S <= 2048:
        for cfg in (0, 1, 2):
            with option("ATTN_QKCFG", cfg):
                close(ops().hypersphere_attention(*args, **kw), ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("kv_bf16", [False, True, "f16keys", "f16scores"])
@pytest.mark.parametrize("B,Lq,S,masked", [(2, 100, 300, True), (2, 100, 100, False), (1, 100, 4800, True), (2, 20, 37, True), (1, 300, 1200, True),
                                           (1, 300, 19200, True)])
def test_hypersphere_attention_low_precision(B, Lq, S, masked, kv_bf16):
    """msm_hypersphere_attn_lp_fwd (bf16 MFMA operands, fp32 accumulation; K / V stored as fp32 or bf16) against the oracle.
    The unit vectors q^, k^ carry 8 mantissa bits, so a logit kappa q^.k^ moves by ~kappa 2^-9 / sqrt(32) ~ 1e-2 and the outputs
    (components of unit vectors) by a few 1e-3; the oracle is fed the bf16-rounded K / V when those are what is stored.
    "f16keys" (precision "f16", kv_format 2): K stored as IEEE half, V as bf16, q^ / k^ on fp16 MFMAs; "f16scores" (kv_format 3): fp32
    K / V with the fp16 score operands -- the scores' error drops eightfold, what is left is the bf16 rounding of the probabilities
    and of V in P V (tighter bounds below)."""
    H, E = 8, 256
    q, k, v = rnd(B, Lq, E, seed=1), rnd(B, S, E, seed=2), rnd(B, S, E, seed=3)
    f16keys, f16scores = kv_bf16 == "f16keys", kv_bf16 == "f16scores"
    if f16keys:
        k, v = k.to(torch.float16).float(), _bf16_round(v)
    elif kv_bf16 is True:
        k, v = _bf16_round(k), _bf16_round(v)
    m = row_any = add = None
    if masked:
        g = torch.Generator().manual_seed(4)
        m = torch.rand(B, Lq, S, generator=g) < 0.6
        m[0, 1] = True
        row_any = (~m.all(-1)).to(torch.int32)
        eff = m.clone()
        eff[m.all(-1)] = False
        add = torch.zeros(B, 1, Lq, S)
        add[eff[:, None]] = float("-inf")
        add = add.expand(B, H, Lq, S).reshape(B * H, Lq, S)
    hd = lambda t, n: t.view(B, n, H, 32).permute(0, 2, 1, 3).reshape(B * H, n, 32)
    o, _ = O.hypersphere_attention(hd(q, Lq), hd(k, S), hd(v, S), add)
    ref = o.view(B, H, Lq, 32).permute(0, 2, 1, 3).reshape(B, Lq, E)
    if f16keys:          # the K columns hold half bit patterns inside a bfloat16-typed tensor (kv_project_multi(keys_f16=True))
        kd, vd = k.to(DEV).to(torch.float16).view(torch.bfloat16), v.to(DEV).to(torch.bfloat16)
    elif kv_bf16 is True:
        kd, vd = k.to(DEV).to(torch.bfloat16), v.to(DEV).to(torch.bfloat16)
    else:
        kd, vd = k.to(DEV), v.to(DEV)
    kw = dict(masked=None if m is None else m.to(torch.uint8).to(DEV), row_any=None if row_any is None else row_any.to(DEV),
              keys_f16=f16keys or f16scores)
    got = ops().hypersphere_attention(q.to(DEV), kd, vd, H, low_precision=True, **kw)
    err = (got.cpu() - ref).abs()
    print(f"lp attention S={S} kv_bf16={kv_bf16}: max |d| {float(err.max()):.2e} mean {float(err.mean()):.2e}")
    if f16keys or f16scores:
        assert float(err.max()) < 1.5e-2 and float(err.mean()) < 8e-4
    assert float(err.max()) < 3e-2 and float(err.mean()) < 2e-3
    nrm = got.view(B, Lq, H, 32).norm(dim=-1)
    close(nrm, torch.ones_like(nrm).cpu(), rtol=1e-5, atol=1e-5)                  # the output normalisation is fp32
    with option_ctx("ATTN_KERNEL", 3):                                           # split-K kernel + combine at every length
        alt = ops().hypersphere_attention(q.to(DEV), kd, vd, H, low_precision=True, **kw)
    assert float((alt.cpu() - ref).abs().max()) < 3e-2


@pytest.mark.parametrize("B,h,w,H,W", [(2, 6, 8, 48, 64), (1, 60, 80, 480, 640), (3, 5, 7, 37, 50), (1, 1, 1, 8, 8)])
def test_ucn_embedding_tail(B, h, w, H, W):
    """msm_ucn_embedding_tail: upsample_bilinear (align_corners=True) of both towers' low-resolution maps, add fusion and the channel
    normalisation(s) of lib/networks/SEG.py:97-117 / pretrained_meanshiftformer_model.py:298-300 in one pass, against the torch ops in
    float64 (one tower, two towers; 0, 1, 2 normalisations)."""
    a = rnd(B, 64, h, w, seed=1).contiguous(memory_format=torch.channels_last)
    b = rnd(B, 64, h, w, seed=2).contiguous(memory_format=torch.channels_last)
    up = lambda t: F.interpolate(t.double(), size=(H, W), mode="bilinear", align_corners=True)
    for two in (False, True):
        for norms in (0, 1, 2):
            ref = up(a) + (up(b) if two else 0)
            for _ in range(norms):
                ref = F.normalize(ref, p=2, dim=1)
            got = ops().ucn_embedding_tail(a.to(DEV), b.to(DEV) if two else None, (H, W), norms=norms)
            assert got.shape == (B, 64, H, W) and got.is_contiguous() and got.dtype == torch.float32
            # (the source coordinate scale * x is an fp32 product, as in at::native's kernel: a tap weight is good to ~1e-5 at x = 639)
            close(got, ref.float(), rtol=1e-4, atol=5e-5)
            # ... and torch's own fp32 kernels on the device give the same map
            dev_ref = F.interpolate(a.to(DEV), size=(H, W), mode="bilinear", align_corners=True)
            if two:
                dev_ref = dev_ref + F.interpolate(b.to(DEV), size=(H, W), mode="bilinear", align_corners=True)
            for _ in range(norms):
                dev_ref = F.normalize(dev_ref, p=2, dim=1)
            close(got, dev_ref.cpu(), rtol=1e-4, atol=5e-5)
    # a map that is not channels_last is taken as well (copied once)
    got2 = ops().ucn_embedding_tail(a.contiguous().to(DEV), None, (H, W), norms=1)
    close(got2, F.normalize(up(a), p=2, dim=1).float(), rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,C,H,W", [(2, 64, 9, 13), (1, 256, 30, 40), (3, 2048, 2, 3), (1, 8, 1, 1)])
def test_backbone_glue_kernels(B, C, H, W, dtype):
    """msm_bias_act_nhwc / msm_nhwc_to_nchw_f32 (csrc/backbone_ops.hip): x = act(x + bias (+ residual)) in place on a channels_last map
    (fp32, bf16, fp16), in fp32 arithmetic with one rounding to the map's dtype -- against float64 rounded once; the NCHW fp32 hand-over exact."""
    x = rnd(B, C, H, W, seed=1).to(dtype).contiguous(memory_format=torch.channels_last)
    r = rnd(B, C, H, W, seed=2).to(dtype).contiguous(memory_format=torch.channels_last)
    b = rnd(C, seed=3).to(dtype)
    for res in (None, r):
        for relu in (True, False):
            ref = x.double() + b.double()[None, :, None, None] + (0 if res is None else res.double())
            ref = ref.clamp_min(0) if relu else ref
            xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
            got = ops().bias_act_nhwc_(xd, b.to(DEV), None if res is None else res.to(DEV).contiguous(memory_format=torch.channels_last), relu)
            assert got.data_ptr() == xd.data_ptr() and got.is_contiguous(memory_format=torch.channels_last)
            if dtype == torch.float32:
                close(got, ref.float(), rtol=1e-6, atol=1e-6)
            else:
                # one rounding of the exact sum: at most one bf16 ulp from the double result rounded to bf16 (fp32 accumulation in between)
                want = ref.to(dtype)
                d = (got.cpu().float() - want.float()).abs()
                ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
                assert float((d > 0).float().mean()) < 0.01 and bool((d <= want.float().abs() * ulp + 1e-7).all())
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    planes = ops().nhwc_to_nchw_f32(xd)
    assert planes.is_contiguous() and planes.dtype == torch.float32 and torch.equal(planes.cpu(), x.float().contiguous())
    with pytest.raises(RuntimeError):
        ops().bias_act_nhwc_(x.to(DEV).contiguous(), b.to(DEV)) if (H * W > 1 and C > 1) else (_ for _ in ()).throw(RuntimeError("n/a"))
    with pytest.raises(RuntimeError):
        ops().bias_act_nhwc_(xd, b.to(DEV)[: C // 2].contiguous())


@pytest.mark.parametrize("B,Q,H,W", [(2, 100, 24, 48), (1, 20, 7, 16), (1, 112, 33, 80), (3, 37, 12, 160)])
def test_mask_conv3x3_folded(B, Q, H, W):
    """msm_mask_conv3x3_folded (UCN path, 16-bit plans): mask = einsum(e, Conv3x3(64 -> 256, padding 1)(x)) (fpn.py:238-246, DEC:1012-1035)
    with the convolution folded into per-query 3x3 filters F = e W.  Against the literal order in float64 -- convolution first, then
    the contraction -- on the operands as the kernel rounds them (x and F to IEEE half); the attention-mask bits (logit < 0), bit-packed
    and blocked as the fused K/V attention reads them, must be those of the float64 logits wherever the logit is not within the
    accumulation error of zero, and row_any must say which rows keep an unmasked key."""
    Cm = 256
    x = F.normalize(rnd(B, 64, H, W, seed=1), dim=1)
    w = rnd(Cm, 64, 3, 3, seed=2, scale=1.0 / 24)
    bias = rnd(Cm, seed=3, scale=0.1)
    e = rnd(B, Q, Cm, seed=4)
    e[0, 1] = 0.0                                      # a query whose logits are all exactly zero (undecided everywhere) ...
    d = lambda t: t.to(DEV).contiguous()
    wf = ops().mask_conv_fold_weight(d(w), d(bias))
    assert wf.shape == (580, Cm)
    # the documented row order: k = 64 * (3 ky + kx) + c, row 576 = bias
    assert torch.equal(wf[:576].cpu(), w.permute(2, 3, 1, 0).reshape(576, Cm)) and torch.equal(wf[576].cpu(), bias) and not wf[577:].any()
    Fq = ops().gemm(d(e), wf)                          # (B, Q, 580) fp32
    Fq[0, 2, 576] = -1e4                               # ... and one that is masked everywhere (row_any = 0)
    Fq[0, 2, :576] = 0.0
    xh = ops().tokens_f16(d(x))
    # float64 reference on the rounded operands: per-query filters (Q, 64, 3, 3) from F, convolution with zero padding
    Fr = Fq.cpu()[..., :576].to(torch.float16).double().view(B, Q, 3, 3, 64).permute(0, 1, 4, 2, 3)
    ref = torch.stack([F.conv2d(xh[b].cpu().double().t().reshape(1, 64, H, W), Fr[b], padding=1)[0] for b in range(B)])
    ref = ref + Fq.cpu()[..., 576].double()[..., None, None]
    # ... which is the literal order up to the fp16 rounding of F: convolution to 256 channels first, then the contraction
    lit = torch.einsum("bqo,bohw->bqhw", e.double(), F.conv2d(x.double(), w.double(), bias.double(), padding=1))
    sel = torch.ones(B, Q, dtype=torch.bool)
    sel[0, 2] = False
    sel[0, 1] = False
    assert float((ref - lit)[sel].abs().max()) < 3e-3 * float(lit.abs().max())
    got = ops().mask_conv3x3_folded(xh, Fq, (H, W), bits=False)
    assert got.shape == (B, Q, H, W)
    tol = 2e-5 * float(ref[sel].abs().max()) + 1e-6
    err = (got.cpu().double() - ref).abs()
    assert float(err[sel].max()) < tol, float(err[sel].max())
    assert float(err[0, 2].max()) < 1e-2                                        # (fp32 spacing at 1e4)
    bits, row_any = ops().mask_conv3x3_folded(xh, Fq, (H, W), bits=True)
    S = H * W
    assert bits.shape == (B, 1, S // 16, 16, 8) and row_any.shape == (B, Q)
    words = bits.cpu().to(torch.int32) & 0xffff                                  # [b, 0, kb, lj, m]: bit k = masked[b, 16 m + lj, 16 kb + k]
    k = torch.arange(16)
    unpacked = ((words[:, 0, :, :, :, None] >> k) & 1).bool()                    # (B, kb, lj, m, k)
    unpacked = unpacked.permute(0, 3, 2, 1, 4).reshape(B, 128, S)[:, :Q]         # query 16 m + lj, key 16 kb + k
    assert not (words[:, 0, :, :, 7] != 0).any()                                 # queries >= 112 never exist
    if Q % 16:
        qpad = ((words[:, 0, :, :, :, None] >> k) & 1).bool().permute(0, 3, 2, 1, 4).reshape(B, 128, S)[:, Q:]
        assert not qpad.any()
    want = ref.view(B, Q, S) < 0
    decided = ref.view(B, Q, S).abs() > tol
    assert torch.equal(unpacked[decided], want[decided])
    assert float(decided[sel].float().mean()) > 0.999 and not decided[0, 1].any()
    ra = row_any.cpu().bool()
    sure_any = ((~want) & decided).any(-1)                                       # some key certainly unmasked
    sure_none = (want & decided).all(-1)                                         # every key certainly masked
    assert ra[sure_any].all() and not ra[sure_none].any()
    assert not ra[0, 2] and ra[0, 1]                                             # (a logit of exactly 0 is not masked: sigmoid(0) < 0.5 is false)
    assert not unpacked[0, 1].any() and unpacked[0, 2].all()
    # a row_any buffer the caller cleared is used as it is
    pre = torch.zeros(B, Q, device=DEV, dtype=torch.int32)
    bits2, ra2 = ops().mask_conv3x3_folded(xh, Fq, (H, W), bits=True, row_any=pre)
    assert ra2.data_ptr() == pre.data_ptr() and torch.equal(bits2, bits) and torch.equal(ra2, row_any)
    with pytest.raises(RuntimeError):
        ops().mask_conv3x3_folded(xh[:, :S - 16].contiguous(), Fq, (H, W))
    if W == 16:
        with pytest.raises(RuntimeError):
            ops().mask_conv3x3_folded(xh.view(B, H * 2, 8, 64).reshape(B, S, 64), Fq, (H * 2, 8))


@pytest.mark.parametrize("keys_f16", [False, True])
@pytest.mark.parametrize("B,Lq,H,W,masked", [(2, 100, 16, 32, True), (1, 100, 40, 160, True), (1, 300, 120, 160, True), (2, 37, 8, 16, False)])
def test_hypersphere_attention_fused_kv(B, Lq, H, W, masked, keys_f16):
    """msm_hypersphere_attn_fused_kv_fwd (16-bit plans, long levels): the folded K/V projection [K | V] = x W^T + row[y] + col[x]
    computed inside the attention kernel from the fp16 level feature.  Against (i) the oracle's hypersphere attention on K / V evaluated
    in float64 from the operands as the kernel rounds them (x and W to fp16) -- the bounds of the unfused low-precision kernel --, and
    (ii) the unfused pair it replaces (msm_kv_project_multi_bf16 + msm_hypersphere_attn_lp_fwd): statistically the same distance from
    the exact result, the fused form a little closer (K is never stored, so it is rounded once less)."""
    Hh, E = 8, 256
    S = H * W
    q = rnd(B, Lq, E, seed=1)
    x = F.normalize(rnd(B, 64, H, W, seed=2), dim=1)                       # a unit-norm embedding map, NCHW like the UCN feature
    w = rnd(2 * E, 64, seed=3, scale=0.35)
    rowcol = rnd(H + W, 2 * E, seed=4, scale=0.3)
    rowcol[H:, E:] *= 0.5
    h16 = lambda t: t.to(torch.float16).double()
    xt = x.flatten(2).transpose(1, 2)                                      # (B, S, 64)
    const = (rowcol[:H, None] + rowcol[None, H:]).reshape(S, 2 * E).double()
    kv = (h16(xt) @ h16(w).t() + const).float()
    k, v = kv[..., :E], kv[..., E:]
    m = row_any = add = None
    if masked:
        g = torch.Generator().manual_seed(4)
        m = torch.rand(B, Lq, S, generator=g) < 0.6
        m[0, 1] = True
        row_any = (~m.all(-1)).to(torch.int32)
        eff = m.clone()
        eff[m.all(-1)] = False
        add = torch.zeros(B, 1, Lq, S)
        add[eff[:, None]] = float("-inf")
        add = add.expand(B, Hh, Lq, S).reshape(B * Hh, Lq, S)
    hd = lambda t, n: t.view(B, n, Hh, 32).permute(0, 2, 1, 3).reshape(B * Hh, n, 32)
    o, _ = O.hypersphere_attention(hd(q, Lq), hd(k, S), hd(v, S), add)
    ref = o.view(B, Hh, Lq, 32).permute(0, 2, 1, 3).reshape(B, Lq, E)
    d = lambda t: t.to(DEV).contiguous()
    xh = ops().tokens_f16(d(x))
    assert xh.shape == (B, S, 64) and torch.equal(xh.cpu(), xt.to(torch.float16))
    # a token-major view of a wider buffer (how the pixel decoder hands its levels over) packs to the same tokens
    buf = torch.zeros(B, S + 5, 64, device=DEV)
    buf[:, 2:2 + S] = d(xt)
    assert torch.equal(ops().tokens_f16(buf[:, 2:2 + S].view(B, H, W, 64).permute(0, 3, 1, 2)), xh)
    wp = ops().attn_pack_kv_weights(d(w), Hh)
    # the documented fragment order: [head][kv][tile t][k-step s][lane (i = l & 15, kq = l >> 4)][8] = W[kv E + 32 h + 16 t + i][32 s + 8 kq ..]
    want = w.view(2, Hh, 2, 16, 2, 4, 8).permute(1, 0, 2, 4, 5, 3, 6).reshape(Hh, 8, 64, 8).to(torch.float16)
    assert torch.equal(wp.cpu(), want)
    if m is not None:
        # the bit-packed, blocked mask: word [b, qc, kb, lj, mb] bit k = masked[b, 112 qc + 16 mb + lj, 16 kb + k]
        bits = ops().attn_pack_mask_bits(d(m.to(torch.uint8))).cpu().to(torch.int32) & 0xffff
        nqc = (Lq + 111) // 112
        mp = torch.zeros(B, nqc * 128, S, dtype=torch.int32)
        for qc in range(nqc):                                             # chunk qc holds queries 112 qc .. (7 blocks of 16), slot 7 empty
            n = min(112, Lq - 112 * qc)
            mp[:, 128 * qc:128 * qc + n] = m[:, 112 * qc:112 * qc + n].to(torch.int32)
        wbits = (mp.view(B, nqc, 8, 16, S // 16, 16) << torch.arange(16, dtype=torch.int32)).sum(-1).permute(0, 1, 4, 3, 2)
        assert torch.equal(bits, wbits)
    kw = dict(masked=None if m is None else d(m.to(torch.uint8)), row_any=None if row_any is None else d(row_any))
    got = ops().hypersphere_attention_fused_kv(d(q), xh, wp, d(rowcol), d(rowcol[H:, E:].t()), (H, W), Hh, keys_f16=keys_f16, **kw)
    err = (got.cpu() - ref).abs()
    print(f"fused K/V attention S={S} keys_f16={keys_f16}: max |d| {float(err.max()):.2e} mean {float(err.mean()):.2e}")
    if keys_f16:
        assert float(err.max()) < 1.5e-2 and float(err.mean()) < 8e-4
    assert float(err.max()) < 3e-2 and float(err.mean()) < 2e-3
    nrm = got.view(B, Lq, Hh, 32).norm(dim=-1)
    close(nrm, torch.ones_like(nrm).cpu(), rtol=1e-5, atol=1e-5)
    # the unfused pair on the same inputs
    kvu = ops().kv_project_multi([d(x)], [d(w)], [d(rowcol)], out_dtype=torch.bfloat16, cmat_widths=[W], keys_f16=keys_f16)[0]
    unf = ops().hypersphere_attention(d(q), kvu[..., :E], kvu[..., E:], Hh, low_precision=True, keys_f16=keys_f16, **kw)
    e_unf = float((unf.cpu() - ref).abs().mean())
    print(f"   unfused pair: mean |d| {e_unf:.2e}")
    assert float(err.mean()) <= 1.25 * e_unf + 1e-5
    with pytest.raises(RuntimeError, match="multiple of 16"):
        ops().hypersphere_attention_fused_kv(d(q), xh[:, :S - H * 8].contiguous(), wp, d(rowcol[:H + W - 8]), d(rowcol[H:H + W - 8, E:].t()), (H, W - 8), Hh)


def test_hypersphere_attention_strided_views():
    B, L, H, E = 2, 100, 8, 256
    qk, v = rnd(B, L, 2 * E, seed=1).to(DEV), rnd(B, L, E, seed=2).to(DEV)
    got = ops().hypersphere_attention(qk[..., :E], qk[..., E:], v, H)
    ref = ops().hypersphere_attention(qk[..., :E].contiguous(), qk[..., E:].contiguous(), v, H)
    assert torch.equal(got, ref)


# ---------------------------------------------------------------------------------------------
def test_msda_reference_known_answer(golden):
    g = golden("msda_core")
    shapes = torch.tensor([(6, 4), (3, 2)], dtype=torch.int64)
    start = torch.tensor([0, 24], dtype=torch.int64)
    T = lambda k: torch.from_numpy(g[k])
    out = ops().ms_deform_attn(T("t_float_value").to(DEV), shapes.to(DEV), start.to(DEV), T("t_float_loc").to(DEV),
                               T("t_float_aw").to(DEV))
    assert torch.allclose(out.cpu(), T("t_float_out"), rtol=1e-2, atol=1e-3)   # ops/test.py:58 criterion
    close(out, T("t_float_out"), rtol=1e-5, atol=1e-8)


def test_msda_realistic_and_fused(golden):
    g = golden("msda_core")
    T = lambda k: torch.from_numpy(g[k])
    shp = [tuple(int(v) for v in r) for r in g["r_shapes"]]
    shapes = torch.tensor(shp, dtype=torch.int64)
    start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    out = ops().ms_deform_attn(T("r_value").to(DEV), shapes.to(DEV), start.to(DEV), T("r_loc").to(DEV), T("r_aw").to(DEV))
    close(out, T("r_out"), rtol=1e-4, atol=1e-5)
    # encoder form: raw offsets + logits, reference points = pixel centres
    N, S, M, D = g["r_value"].shape
    L, P = 3, 4
    proj = rnd(N, S, M * L * P * 3, seed=5)
    proj[..., :M * L * P * 2] *= 3.0
    value = T("r_value").reshape(N, S, M * D)
    off = proj[..., :M * L * P * 2].reshape(N, S, M, L, P, 2)
    aw = torch.softmax(proj[..., M * L * P * 2:].reshape(N, S, M, L * P), -1).reshape(N, S, M, L, P)
    ref_pts = O.encoder_reference_points(shp, N)
    norm = torch.tensor([[w, h] for h, w in shp], dtype=torch.float32)
    loc = ref_pts[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    ref = O.ms_deform_attn_core(T("r_value"), shp, loc, aw)
    got = ops().ms_deform_attn_encoder(value.contiguous().to(DEV), shapes.to(DEV), start.to(DEV), proj.to(DEV), M, P)
    close(got, ref, rtol=1e-4, atol=1e-5)
    # head-major value layout (N, M, S, D): same function, two taps per 64-byte segment
    value_hm = T("r_value").permute(0, 2, 1, 3).contiguous()
    assert torch.equal(ops().value_to_head_major(value.contiguous().to(DEV), M).cpu(), value_hm)
    got_hm = ops().ms_deform_attn_encoder(value_hm.to(DEV), shapes.to(DEV), start.to(DEV), proj.to(DEV), M, P)
    close(got_hm, ref, rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------------------------
def closed(got, ref, rtol, atol):
    torch.testing.assert_close(got.double().cpu(), ref, rtol=rtol, atol=atol)


def _ln(x, g, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), g, b, eps)


@pytest.mark.parametrize("B,Q", [(2, 100), (1, 7), (3, 16)])
@pytest.mark.parametrize("prec", ["f32", "bf16", "f16", "bf16x2"])
def test_decoder_fused_tails(B, Q, prec):
    """csrc/dec_chain.hip against the same chain in torch fp64 (DEC:245-260, 171-181, 296-300, 637-638, 661-665);
    tolerance: fp32 rounding of 256/2048-term dot products on O(1) values.  prec = "bf16": the low-precision entry points
    (bf16 weight fragments, activations as hi + lo bf16 pairs, fp32 accumulation) against the same fp64 chain on the
    bf16-ROUNDED weights -- what is left is the 2^-17 residual of the activation split.  prec = "f16" (round 5): IEEE-half weight
    fragments and ONE fp16 activation term per product, against the fp64 chain with the weights AND each GEMM's input rounded to
    fp16 -- what is left is fp32 accumulation; that this form is closer to the exact chain than the bf16 one is
    test_decoder_tails_f16_is_closer_to_exact_than_bf16.  prec = "bf16x2" (round 6, what the "bf16" plan runs): hi + lo bf16 WEIGHT
    fragments beside the hi + lo activation split -- against the fp64 chain on the EXACT weights under the bf16 form's bound (what is
    left is 2^-17 per operand)."""
    E, Fh = 256, 2048
    r = lambda *s, seed, k=1.0: (rnd(*s, seed=seed) * k)
    o, res, qpos = r(B, Q, E, seed=1), r(B, Q, E, seed=2), r(Q, E, seed=3)
    wo, bo = r(E, E, seed=4, k=E ** -0.5), r(E, seed=5, k=0.1)
    g, b = 1 + r(E, seed=6, k=0.1), r(E, seed=7, k=0.1)
    w_in, b_in = r(3 * E, E, seed=8, k=E ** -0.5), r(3 * E, seed=9, k=0.1)
    dev = lambda *ts: [t.to(DEV) for t in ts]
    dbl = lambda *ts: [t.double() for t in ts]
    bf = prec == "bf16"
    f16 = prec == "f16"
    x2 = prec == "bf16x2"
    pack = {"f32": lambda w: ops().dec_pack_weight(w.to(DEV)), "bf16": lambda w: ops().dec_pack_weight_bf16(w.to(DEV)),
            "f16": lambda w: ops().dec_pack_weight_f16(w.to(DEV)), "bf16x2": lambda w: ops().dec_pack_weight_bf16x2(w.to(DEV))}[prec]
    rw = _bf16_round if bf else ((lambda w: w.to(torch.float16).float()) if f16 else (lambda w: w))      # the weights the kernels actually multiply by
    # (f16: a stage's input differs from the reference's by ~1e-5, so a few per cent of its elements round to the neighbouring half --
    # each such flip is 2^-11 |x| |w|: the price of rounding activations at all, and why this form's bound is three times the bf16 form's)
    tol = 12.0 if f16 else (4.0 if (bf or x2) else 1.0)
    # f16: the activation enters every product as ONE fp16 term -- the references round it the same way in front of each GEMM
    A = (lambda t: t.float().to(torch.float16).double()) if f16 else (lambda t: t)
    closed_ = globals()["closed"]
    closed = lambda got, ref, rtol, atol: closed_(got, ref, rtol=rtol * tol, atol=atol * tol + (4e-5 if f16 else 0.0))  # noqa: E731
    # the documented fragment order (include/msm_hip.h)
    N_, K_ = w_in.shape
    frag = lambda m, dt: m.view(N_ // 16, 16, K_ // 64, 2, 2, 4, 4).permute(0, 2, 3, 5, 1, 4, 6).contiguous().view(N_ // 16, K_ // 64, 1024).to(dt)
    if x2:              # per row tile: the K / 64 chunks of bf16(W), then the K / 64 chunks of bf16(W - bf16(W))
        hi = w_in.to(torch.bfloat16)
        want = torch.cat([frag(hi.float(), torch.bfloat16), frag(w_in - hi.float(), torch.bfloat16)], 1).reshape(N_, 2 * K_)
    elif bf or f16:     # [t][kc][up][lq][lj][h][c] <- W[t*16 + lj][kc*64 + (2 up + h)*16 + lq*4 + c]
        want = frag(w_in, torch.float16 if f16 else torch.bfloat16).reshape(N_, K_)
    else:
        want = w_in.view(N_ // 16, 16, K_ // 64, 4, 4, 4).permute(0, 2, 3, 4, 1, 5).contiguous().view(N_, K_)
    assert torch.equal(pack(w_in).cpu(), want)
    wo, w_in = rw(wo), rw(w_in)                              # references below use the rounded weights; pack() rounds again (idempotent)
    # post_cross
    x, qk, v = ops().dec_post_cross(*dev(o, res, qpos), pack(wo), *dev(bo, g, b), pack(w_in), b_in.to(DEV))
    O_, R_, P_, WO, BO, G_, B_, WI, BI = dbl(o, res, qpos, wo, bo, g, b, w_in, b_in)
    xr = _ln(R_ + A(O_) @ WO.t() + BO, G_, B_)
    closed(x, xr, rtol=1e-4, atol=2e-5)
    closed(qk, A(xr + P_) @ WI[:2 * E].t() + BI[:2 * E], rtol=1e-4, atol=5e-5)
    closed(v, A(xr) @ WI[2 * E:].t() + BI[2 * E:], rtol=1e-4, atol=5e-5)
    # post_self
    w1, b1 = rw(r(Fh, E, seed=10, k=E ** -0.5)), r(Fh, seed=11, k=0.1)
    w2, b2 = rw(r(E, Fh, seed=12, k=Fh ** -0.5)), r(E, seed=13, k=0.1)
    x2, parts = ops().dec_post_self(*dev(o, res), pack(wo), *dev(bo, g, b), pack(w1), b1.to(DEV), pack(w2))
    closed(x2, xr, rtol=1e-4, atol=2e-5)
    W1, B1, W2, B2 = dbl(w1, b1, w2, b2)
    ffn = A(torch.relu(A(xr) @ W1.t() + B1)) @ W2.t()
    closed(parts.double().sum(0), ffn, rtol=1e-4, atol=5e-5)
    for n_parts in (1, 2, 8):
        x3, p3 = ops().dec_post_self(*dev(o, res), pack(wo), *dev(bo, g, b), pack(w1), b1.to(DEV), pack(w2), n_parts=n_parts)
        assert p3.shape == (n_parts, B, Q, E) and torch.equal(x3, x2)
        closed(p3.double().sum(0), ffn, rtol=1e-4, atol=5e-5)
    with pytest.raises(RuntimeError, match="must divide"):
        ops().dec_post_self(*dev(o, res), pack(wo), *dev(bo, g, b), pack(w1), b1.to(DEV), pack(w2), n_parts=3)
    # heads (with and without the optional pieces)
    g1, be1 = 1 + r(E, seed=14, k=0.1), r(E, seed=15, k=0.1)
    g2, be2 = 1 + r(E, seed=16, k=0.1), r(E, seed=17, k=0.1)
    mlp = [(rw(r(E, E, seed=20 + i, k=E ** -0.5)), r(E, seed=30 + i, k=0.1)) for i in range(3)]
    wq, bq = rw(r(E, E, seed=40, k=E ** -0.5)), r(E, seed=41, k=0.1)
    mlp_d = [(pack(w), bb.to(DEV)) for w, bb in mlp]
    out, d, e, q = ops().dec_heads(x2, *dev(g2, be2), mlp_d, parts=parts, bias=b2.to(DEV), ln_g=g1.to(DEV), ln_b=be1.to(DEV),
                                   l2norm=True, wq=pack(wq), bq=bq.to(DEV), query_pos=qpos.to(DEV), want_d=True)
    t = _ln(xr + ffn + B2, g1.double(), be1.double())
    t = t / t.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    dr = _ln(t, g2.double(), be2.double())
    er = dr
    for i, (w, bb) in enumerate(mlp):
        er = A(er) @ w.double().t() + bb.double()
        if i < 2:
            er = torch.relu(er)
    closed(out, t, rtol=1e-4, atol=1e-6)
    closed(d, dr, rtol=1e-4, atol=5e-5)
    closed(e, er, rtol=1e-4, atol=1e-4)
    closed(q, A(t + P_) @ wq.double().t() + bq.double(), rtol=1e-4, atol=5e-5)
    # initial form: no parts, no FFN norm, no block norm, nothing optional
    out0, d0, e0, q0 = ops().dec_heads(res.to(DEV), *dev(g2, be2), mlp_d, want_out=False)
    assert out0 is None and d0 is None and q0 is None
    er = _ln(R_, g2.double(), be2.double())
    for i, (w, bb) in enumerate(mlp):
        er = A(er) @ w.double().t() + bb.double()
        if i < 2:
            er = torch.relu(er)
    closed(e0, er, rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError, match="all float32, all bfloat16 or all float16"):
        ops().dec_post_cross(*dev(o, res, qpos), ops().dec_pack_weight_bf16(wo.to(DEV)), *dev(bo, g, b), ops().dec_pack_weight_f16(w_in.to(DEV)), b_in.to(DEV))
    with pytest.raises(RuntimeError, match="all hi \\+ lo bf16 fragments"):
        ops().dec_post_cross(*dev(o, res, qpos), ops().dec_pack_weight_bf16(wo.to(DEV)), *dev(bo, g, b), ops().dec_pack_weight_bf16x2(w_in.to(DEV)), b_in.to(DEV))


def test_decoder_tails_f16_is_closer_to_exact_than_bf16():
    """Why precision "f16" exists: the same FFN stage (msm_dec_post_self) against the fp64 chain on the EXACT weights -- the fp16 form
    (2^-12 roundings of weights and activations) must be at least four times closer than the bf16 form (2^-9 on the weights)."""
    B, Q, E, Fh = 2, 100, 256, 2048
    r = lambda *s_, seed, k=1.0: (rnd(*s_, seed=seed) * k)
    o, res = r(B, Q, E, seed=1), r(B, Q, E, seed=2)
    wo, bo, g, b = r(E, E, seed=4, k=E ** -0.5), r(E, seed=5, k=0.1), 1 + r(E, seed=6, k=0.1), r(E, seed=7, k=0.1)
    w1, b1, w2 = r(Fh, E, seed=10, k=E ** -0.5), r(Fh, seed=11, k=0.1), r(E, Fh, seed=12, k=Fh ** -0.5)
    xr = _ln(res.double() + o.double() @ wo.double().t() + bo.double(), g.double(), b.double())
    ffn = torch.relu(xr @ w1.double().t() + b1.double()) @ w2.double().t()
    errs = {}
    for prec, pk in (("bf16", ops().dec_pack_weight_bf16), ("f16", ops().dec_pack_weight_f16), ("f32", ops().dec_pack_weight),
                     ("bf16x2", ops().dec_pack_weight_bf16x2)):
        d = lambda t: t.to(DEV)
        _, parts = ops().dec_post_self(d(o), d(res), pk(d(wo)), d(bo), d(g), d(b), pk(d(w1)), d(b1), pk(d(w2)))
        errs[prec] = float((parts.double().sum(0).cpu() - ffn).abs().mean())
    print(f"FFN stage, mean |error| against fp64 on the exact weights: {errs}")
    assert errs["f16"] * 4 <= errs["bf16"] and errs["f32"] <= errs["f16"]
    assert errs["bf16x2"] * 8 <= errs["bf16"] and errs["bf16x2"] <= errs["f16"]          # hi + lo weights: the 2^-9 of the weights is gone


def test_decoder_tails_f16_32_row_tiles_equal_16_row_tiles():
    """Round 6, TileQ32 (csrc/dec_chain.hip): above 4096 rows -- the second stage of configs[3] -- the fp16 tails run 32 rows per
    workgroup, two 16-row MFMA tiles sharing every weight fragment.  A row's arithmetic does not depend on the tile it sits in: all
    three kernels must return bit for bit what the 16-row form returns, at row counts with a ragged last tile (rows % 32 in 1..31)."""
    from unseenobjectswithmeanshift_amd import _lib
    E, Fh = 256, 2048
    r = lambda *s_, seed, k=1.0: (rnd(*s_, seed=seed) * k).to(DEV)
    pk = ops().dec_pack_weight_f16
    wo, bo, g, b = pk(r(E, E, seed=4, k=E ** -0.5)), r(E, seed=5, k=0.1), 1 + r(E, seed=6, k=0.1), r(E, seed=7, k=0.1)
    w_in, b_in = pk(r(3 * E, E, seed=8, k=E ** -0.5)), r(3 * E, seed=9, k=0.1)
    w1, b1, w2, b2 = pk(r(Fh, E, seed=10, k=E ** -0.5)), r(Fh, seed=11, k=0.1), pk(r(E, Fh, seed=12, k=Fh ** -0.5)), r(E, seed=13, k=0.1)
    g1, be1, g2, be2 = 1 + r(E, seed=14, k=0.1), r(E, seed=15, k=0.1), 1 + r(E, seed=16, k=0.1), r(E, seed=17, k=0.1)
    mlp = [(pk(r(E, E, seed=20 + i, k=E ** -0.5)), r(E, seed=30 + i, k=0.1)) for i in range(3)]
    wq, bq = pk(r(E, E, seed=40, k=E ** -0.5)), r(E, seed=41, k=0.1)
    for B, Q in ((3, 7), (2, 100), (45, 100)):                 # 21, 200 and 4500 rows (the last one takes TileQ32 by default)
        o, res, qpos = r(B, Q, E, seed=1), r(B, Q, E, seed=2), r(Q, E, seed=3)
        got = {}
        for tile32 in (0, 1):
            with _lib.option("DEC_TILE32", tile32):
                x, qk, v = ops().dec_post_cross(o, res, qpos, wo, bo, g, b, w_in, b_in)
                outs = [x, qk, v]
                for n_parts in (1, 4):
                    x2, parts = ops().dec_post_self(o, res, wo, bo, g, b, w1, b1, w2, n_parts=n_parts)
                    out, d, e, q, ra = ops().dec_heads(x2, g2, be2, mlp, parts=parts, bias=b2, ln_g=g1, ln_b=be1, l2norm=True, wq=wq, bq=bq,
                                                       query_pos=qpos, want_d=True, zero_row_any=True)
                    outs += [x2, parts, out, d, e, q, ra]
                got[tile32] = outs
        for a, c in zip(got[0], got[1]):
            assert torch.equal(a, c)
        if B * Q >= 4096:                                       # the default choice at this size is the 32-row form
            x2, parts = ops().dec_post_self(o, res, wo, bo, g, b, w1, b1, w2, n_parts=1)
            assert torch.equal(parts, got[1][4])


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_decoder_heads_with_mask_epilogue(prec):
    """Round 6, msm_dec_heads_mask: the heads kernel with the NEXT layer's attention mask at key resolution as its epilogue (image-aligned
    tiles, the key blocks of an image shared out over several workgroups per tile) returns bit for bit what the two launches
    dec_heads + attn_mask_pooled return -- out / e / next query, the mask (bytes, and the bit-packed blocked form the fused K/V attention
    reads) and the row flags -- in both operand forms of the mask contraction, for ragged last tiles (Q = 100, 7) and key counts with a
    ragged last block."""
    E = 256
    r = lambda *s_, seed, k=1.0: (rnd(*s_, seed=seed) * k).to(DEV)
    pk = ops().dec_pack_weight_bf16 if prec == "bf16" else ops().dec_pack_weight_f16
    g1, be1, g2, be2 = 1 + r(E, seed=14, k=0.1), r(E, seed=15, k=0.1), 1 + r(E, seed=16, k=0.1), r(E, seed=17, k=0.1)
    mlp = [(pk(r(E, E, seed=20 + i, k=E ** -0.5)), r(E, seed=30 + i, k=0.1)) for i in range(3)]
    wq, bq, b2 = pk(r(E, E, seed=40, k=E ** -0.5)), r(E, seed=41, k=0.1), r(E, seed=13, k=0.1)
    for B, Q, T in ((3, 100, 300), (2, 100, 1200), (2, 100, 4800), (2, 7, 77), (1, 16, 48), (8, 100, 1200)):
        x, qpos = r(B, Q, E, seed=1), r(Q, E, seed=3)
        parts = r(4, B, Q, E, seed=2, k=0.3)
        pooled = r(B, T, 64, seed=5)
        for f16ops in (False, True):
            for bits in ((False, True) if T % 16 == 0 else (False,)):
                for with_q in (True, False):
                    kw = dict(parts=parts, bias=b2, ln_g=g1, ln_b=be1, l2norm=True)
                    if with_q:
                        kw.update(wq=wq, bq=bq, query_pos=qpos)
                    out, d, e, q, ra = ops().dec_heads(x, g2, be2, mlp, want_d=True, zero_row_any=True, **kw)
                    attn, ra = ops().attn_mask_pooled(e[..., :64], pooled, qbias=e[..., 64], row_any=ra, bits=bits, f16=f16ops)
                    flags = torch.zeros((B, Q), dtype=torch.int32, device=DEV)
                    out2, d2, e2, q2, attn2, ra2 = ops().dec_heads_mask(x, g2, be2, mlp, pooled, flags, qcol=64, bits=bits, f16=f16ops, want_d=True, **kw)
                    tag = f"B={B} Q={Q} T={T} f16ops={f16ops} bits={bits} q={with_q}"
                    assert torch.equal(out, out2) and torch.equal(d, d2) and torch.equal(e, e2), tag
                    assert (q is None and q2 is None) or torch.equal(q, q2), tag
                    assert torch.equal(ra, ra2), tag
                    if bits:
                        # (B, chunks, T / 16, 16 queries of a block, 8 blocks of the chunk): words of queries >= Q are written by neither form
                        lj, mb = torch.meshgrid(torch.arange(16), torch.arange(8), indexing="ij")
                        ok = ((mb * 16 + lj < Q) & (mb < 7)).to(DEV)
                        assert Q <= 112 and torch.equal(attn[:, 0][..., ok], attn2[:, 0][..., ok]), tag
                    else:
                        assert torch.equal(attn, attn2), tag
    with pytest.raises(RuntimeError, match="bf16 or fp16"):
        m32 = [(ops().dec_pack_weight(r(E, E, seed=50 + i)), r(E, seed=60 + i)) for i in range(3)]
        ops().dec_heads_mask(r(1, 16, E, seed=1), g2, be2, m32, r(1, 48, 64, seed=5), torch.zeros((1, 16), dtype=torch.int32, device=DEV))


def test_decoder_fused_tails_reject_bad_sizes():
    E = 128
    z = lambda *s: torch.zeros(*s, device=DEV)
    with pytest.raises(RuntimeError, match="only 256"):
        ops().dec_post_cross(z(1, 4, E), z(1, 4, E), z(4, E), z(E, E), z(E), z(E), z(E), z(3 * E, E), z(3 * E))


@pytest.mark.parametrize("B,H,W,N", [(8, 30, 40, 512), (3, 60, 80, 512), (2, 61, 67, 256), (1, 15, 20, 512)])
def test_kv_project(B, H, W, N):
    """msm_kv_project_f32 (and its small-shape GEMM route) vs x^T w^T + cmat in fp64 (DEC:575/251, AU:134-140 folded)."""
    x, w, c = rnd(B, 64, H, W, seed=1), rnd(N, 64, seed=2, scale=0.125), rnd(H * W, N, seed=3)
    got = ops().kv_project(x.to(DEV), w.to(DEV), c.to(DEV))
    ref = torch.einsum("bkp,nk->bpn", x.double().flatten(2), w.double()) + c.double()
    closed(got, ref, rtol=1e-5, atol=2e-5)
    # token-major (channels_last) input inside a wider token buffer, as the pixel decoder hands it over
    buf = torch.zeros(B, H * W + 7, 64, device=DEV)
    buf[:, 3:3 + H * W] = x.flatten(2).transpose(1, 2).to(DEV)
    view = buf[:, 3:3 + H * W].view(B, H, W, 64).permute(0, 3, 1, 2)
    assert ops().is_token_major(view) and not view.is_contiguous()
    closed(ops().kv_project(view, w.to(DEV), c.to(DEV)), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("B,H,W,N", [(8, 30, 40, 512), (2, 61, 67, 256), (1, 15, 20, 512), (2, 96, 128, 512)])
def test_kv_project_separable_constant(B, H, W, N):
    """cmat_width = W: the constant of token (y, x) is row[y] + col[x] (two tables of H + W vectors, what the sine position embedding
    folds to) -- every kernel form (fp32 MFMA, exact three-term splits, bf16) against float64 and against its own dense-constant
    launch; the small-shape GEMM route densifies."""
    x, w = rnd(B, 64, H, W, seed=1), rnd(N, 64, seed=2, scale=0.125)
    rc = rnd(H + W, N, seed=3)
    dense = ops().dense_kv_constant(rc.to(DEV), W)
    assert torch.equal(dense.cpu().view(H, W, N), rc[:H, None] + rc[None, H:])
    ref = torch.einsum("bkp,nk->bpn", x.double().flatten(2), w.double()) + dense.double().cpu()
    xd, wd, cd = x.to(DEV), w.to(DEV), rc.to(DEV)
    got = ops().kv_project(xd, wd, cd, W)
    closed(got, ref, rtol=1e-5, atol=2e-5)
    close(got, ops().kv_project(xd, wd, dense).cpu(), rtol=0, atol=1e-5)            # fl(fl(s + row) + col) against fl(s + fl(row + col)): a few ulps at |v| ~ 8
    for kw, tol in ((dict(), 2e-5), (dict(split=True), 2e-5), (dict(out_dtype=torch.bfloat16), 3e-2)):
        a = ops().kv_project_multi([xd, xd], [wd, wd], [cd, cd], cmat_widths=[W, W], **kw)
        b = ops().kv_project_multi([xd, xd], [wd, wd], [dense, dense], **kw)
        for u, v in zip(a, b):
            closed(u.float(), ref, rtol=tol, atol=tol)
            close(u.float(), v.float().cpu(), rtol=0, atol=1e-5 if u.dtype == torch.float32 else 7e-2)   # (bf16: one rounding step at |v| ~ 8)
    with pytest.raises(RuntimeError):                      # all jobs separable or none
        ops().kv_project_multi([xd, xd], [wd, wd], [cd, dense], cmat_widths=[W, 0])


@pytest.mark.parametrize("B,H,W,N,gn", [(2, 24, 32, 256, True), (3, 10, 14, 256, False), (1, 30, 40, 512, True)])
def test_tokens_proj_nchw(B, H, W, N, gn):
    """msm_tokens_proj_nchw_f32 = 1x1 conv to NCHW of relu(GroupNorm(x)) (MSD:349-358) against torch in fp64."""
    x = rnd(B, H * W, 64, seed=1) * 2 + 0.5
    w, b = rnd(N, 64, seed=2, scale=0.125), rnd(N, seed=3)
    gamma, beta = 1 + rnd(64, seed=4, scale=0.2), rnd(64, seed=5, scale=0.2)
    xd = x.to(DEV)
    if gn:
        got = ops().tokens_proj_nchw(xd, w.to(DEV), b.to(DEV), gn=(ops().groupnorm_stats(xd), gamma.to(DEV), beta.to(DEV), 32, 1e-5),
                                     relu=True)
        y = torch.relu(F.group_norm(x.double().transpose(1, 2), 32, gamma.double(), beta.double(), 1e-5))     # (B, 64, HW)
    else:
        got = ops().tokens_proj_nchw(xd, w.to(DEV), b.to(DEV))
        y = x.double().transpose(1, 2)
    ref = torch.einsum("nk,bkp->bnp", w.double(), y) + b.double()[None, :, None]
    closed(got, ref, rtol=1e-4, atol=5e-5)


from unseenobjectswithmeanshift_amd._lib import option as option_ctx  # noqa: E402


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("B,Q,C,H,W,tgt", [(2, 100, 256, 24, 32, (12, 16)), (1, 37, 64, 16, 48, (4, 12)), (2, 100, 256, 24, 32, (3, 4)),
                                           (1, 120, 128, 8, 16, (8, 16)), (1, 20, 256, 24, 32, None)])
def test_mask_logits_bf16(B, Q, C, H, W, tgt):
    """bf16 mask step (msm_mask_logits_bf16_fwd): logits equal the einsum of the bf16-ROUNDED operands accumulated in
    fp32 (rtol 1e-4: only the summation order differs); attention bits are the sign of the 2x2 tap sums of those logits."""
    e, f = rnd(B, Q, C, seed=1, scale=0.3), rnd(B, C, H, W, seed=2)
    packed = ops().pack_mask_features_bf16(f.to(DEV))
    # the packed layout is (B, C/4, HW, 4) bf16
    want = f.view(B, C // 4, 4, H * W).permute(0, 1, 3, 2).to(torch.bfloat16).contiguous()
    assert torch.equal(packed.cpu().view(torch.bfloat16), want)
    mask, attn, row_any = ops().mask_logits(e.to(DEV), f.to(DEV), want_mask=True, target_size=tgt, packed_bf16=packed)
    ref = torch.einsum("bqc,bchw->bqhw", _bf16_round(e).double(), _bf16_round(f).double())
    closed(mask, ref, rtol=1e-4, atol=1e-4)
    if tgt is None:
        assert attn is None
        return
    s = H // tgt[0]
    m = mask.cpu().double()
    if s == 1:
        bits = m < 0
    else:
        r0 = torch.arange(tgt[0]) * s + s // 2 - 1
        c0 = torch.arange(tgt[1]) * s + s // 2 - 1
        tap = (m[:, :, r0][:, :, :, c0] + m[:, :, r0][:, :, :, c0 + 1]) + (m[:, :, r0 + 1][:, :, :, c0] + m[:, :, r0 + 1][:, :, :, c0 + 1])
        bits = tap < 0
    got = attn.cpu().view(B, Q, tgt[0], tgt[1]).bool()
    near = (tap.abs() if s > 1 else m.abs()) < 1e-4
    assert ((got != bits) & ~near).sum() == 0
    assert torch.equal(row_any.cpu().bool(), (~got).flatten(2).any(-1))
    # and against the fp32 step: bf16 operand rounding moves logits by ~1e-2 relative, flips a percent of the bits
    m32, a32, _ = ops().mask_logits(e.to(DEV), f.to(DEV), want_mask=True, target_size=tgt)
    scale = float(m32.abs().max())
    assert float((mask - m32).abs().max()) < 3e-2 * scale
    assert (attn != a32).float().mean() < 0.03
    # precision "f16" (pack_mask_features_bf16(f16=True) -> MSM_MASK_F16): the einsum of the fp16-ROUNDED operands in fp32, and an
    # eighth of the bf16 step's distance from the fp32 step
    packed_h = ops().pack_mask_features_bf16(f.to(DEV), f16=True)
    assert packed_h.dtype == torch.float16
    assert torch.equal(packed_h.cpu(), f.view(B, C // 4, 4, H * W).permute(0, 1, 3, 2).to(torch.float16))
    mh, ah, rah = ops().mask_logits(e.to(DEV), f.to(DEV), want_mask=True, target_size=tgt, packed_bf16=packed_h)
    ref_h = torch.einsum("bqc,bchw->bqhw", e.to(torch.float16).double(), f.to(torch.float16).double())
    close(mh.double(), ref_h, rtol=1e-4, atol=2e-4)
    assert 4 * float((mh - m32).abs().mean()) <= float((mask - m32).abs().mean())
    if tgt is not None:
        assert (ah != a32).float().mean() < 0.005 and torch.equal(rah.cpu().bool(), (~ah.cpu().view(B, Q, -1).bool()).any(-1))


def _start(shapes):
    return torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))


@pytest.mark.parametrize("tag", ["t30", "t32", "t64", "r"])
def test_msda_backward(golden, tag):
    """msm_msdeform_attn_bwd vs fp64 autograd through the reference op (ops/test.py:66-89 recipe; the reference
    accepts rel err 1e-2/abs 1e-3 there, this holds fp32 rounding).  t30: D/4 lanes not a power of two -> atomic path."""
    g = golden("msda_backward")
    G = lambda k: torch.from_numpy(g[f"{tag}_{k}"]).float()
    shapes = torch.tensor(g[f"{tag}_shapes"], dtype=torch.int64)
    gv, gl, gw = ops().ms_deform_attn_backward(G("value").to(DEV), shapes.to(DEV), _start(shapes).to(DEV), G("loc").to(DEV),
                                               G("aw").to(DEV), G("gout").to(DEV))
    scale = lambda t: float(t.abs().max())
    close(gv, G("gvalue"), rtol=1e-4, atol=1e-5 * scale(G("gvalue")))
    close(gl, G("gloc"), rtol=1e-4, atol=1e-5 * scale(G("gloc")))
    close(gw, G("gaw"), rtol=1e-4, atol=1e-5 * scale(G("gaw")))


def test_msda_module_shim_autograd(golden):
    """The reference's MSDeformAttnFunction pattern (ms_deform_attn_func.py:32-49) over the drop-in module."""
    from unseenobjectswithmeanshift_amd import MultiScaleDeformableAttention as MSDA

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, value, shapes, start, loc, aw, step):
            ctx.step = step
            ctx.save_for_backward(value, shapes, start, loc, aw)
            return MSDA.ms_deform_attn_forward(value, shapes, start, loc, aw, step)

        @staticmethod
        def backward(ctx, go):
            gv, gl, gw = MSDA.ms_deform_attn_backward(*ctx.saved_tensors, go.contiguous(), ctx.step)
            return gv, None, None, gl, gw, None

    g = golden("msda_backward")
    G = lambda k: torch.from_numpy(g[f"r_{k}"]).float()
    shapes = torch.tensor(g["r_shapes"], dtype=torch.int64)
    v, l, a = (G(k).to(DEV).requires_grad_(True) for k in ("value", "loc", "aw"))
    out = Fn.apply(v, shapes.to(DEV), _start(shapes).to(DEV), l, a, 64)
    close(out, G("out"), rtol=1e-4, atol=1e-5)
    out.backward(G("gout").to(DEV))
    close(v.grad, G("gvalue"), rtol=1e-4, atol=1e-4)
    close(l.grad, G("gloc"), rtol=1e-4, atol=1e-3)
    close(a.grad, G("gaw"), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_backward(v.detach(), shapes.to(DEV), _start(shapes).to(DEV), l.detach(), a.detach(),
                                     G("gout").to(DEV)[:, :, ::2], 64)


# ---------------------------------------------------------------------------------------------
def test_mean_shift_kernels(golden):
    from unseenobjectswithmeanshift_amd import synthetic as syn
    g = golden("mean_shift")
    X, _ = syn.synth_unit_embeddings(2000, 64, clusters=6, sigma=0.15, seed=1)
    Xd = X.to(DEV)
    seeds, sel = ops().ms_select_seeds(Xd, 20, int(g["s_first"]))
    assert torch.equal(sel.cpu(), torch.from_numpy(g["s_sel"]))
    assert torch.equal(seeds.cpu(), torch.from_numpy(g["s_seeds"]))
    Z = ops().ms_hill_climb(Xd, seeds, 20.0, 10)
    close(Z, torch.from_numpy(g["s_Z"]), rtol=1e-4, atol=1e-5)
    seed_labels = torch.from_numpy(g["s_cc"])
    num = int(seed_labels.max()) + 1
    labels, counts = ops().ms_assign(Xd, Z, seed_labels.to(DEV), num)
    Zc = Z.cpu()
    ref_lab = seed_labels[torch.argmin(0.5 * (1 - X @ Zc.t()), dim=1)]
    assert (labels.cpu() != ref_lab).float().mean() < 1e-3
    assert torch.equal(counts.cpu(), torch.bincount(labels.cpu(), minlength=num))
    lab2 = ops().ms_relabel_largest_zero(labels.clone(), counts)
    lmax = int(torch.argmax(counts.cpu()))
    exp = labels.cpu().clone()
    if lmax != 0:
        exp[labels.cpu() == 0] = lmax
        exp[labels.cpu() == lmax] = 0
    assert torch.equal(lab2.cpu(), exp)


@pytest.mark.parametrize("n", [4096, 70001, 150000, 393216])
def test_mean_shift_persistent_seeding_equals_stepwise(n, lib_option):
    """The single-launch persistent seeding kernel (map held in registers, grid barrier per step) selects exactly the
    indices of the one-launch-per-step path: same butterfly dot products, same (value, ~index) keys."""
    from unseenobjectswithmeanshift_amd import synthetic as syn
    X, _ = syn.synth_unit_embeddings(n, 64, clusters=9, sigma=0.2, seed=n % 97)
    Xd = X.to(DEV)
    seeds_p, sel_p = ops().ms_select_seeds(Xd, 40, n // 3)
    lib_option("MS_NO_PERSISTENT", 1)
    seeds_s, sel_s = ops().ms_select_seeds(Xd, 40, n // 3)
    assert int(sel_p.min()) >= 0                       # -1 would mean the persistent kernel gave up
    assert torch.equal(sel_p, sel_s) and torch.equal(seeds_p, seeds_s)
    assert int(sel_p[0]) == n // 3 and sel_p.unique().numel() == 40


def test_mean_shift_seeding_give_up_falls_back(monkeypatch):
    """When the persistent seeding kernel loses co-residency it gives up (every index -1); mean_shift_smart_init and
    select_smart_seeds then re-run seeding on the one-launch-per-step path instead of raising -- same labels as a clean run."""
    from unseenobjectswithmeanshift_amd import mean_shift as ms
    from unseenobjectswithmeanshift_amd import ops as opsmod
    from unseenobjectswithmeanshift_amd import synthetic as syn
    X, _ = syn.synth_unit_embeddings(20000, 64, clusters=7, sigma=0.15, seed=5)
    Xd = X.to(DEV)
    _, sel = ops().ms_select_seeds(Xd, 30, 11, _test_give_up=True)
    assert int(sel.max()) == -1                                    # the simulated give-up is reported, not fabricated
    clean_labels, clean_sel = ms.mean_shift_smart_init(Xd, kappa=20, num_seeds=30, max_iters=10, first_index=11)
    real = opsmod.ms_select_seeds
    calls = []

    def flaky(X_, S_, first, stepwise=False, _test_give_up=False, xb=None):
        calls.append(stepwise)
        return real(X_, S_, first, stepwise=stepwise, _test_give_up=not stepwise, xb=xb)

    monkeypatch.setattr(opsmod, "ms_select_seeds", flaky)
    labels, sel = ms.mean_shift_smart_init(Xd, kappa=20, num_seeds=30, max_iters=10, first_index=11)
    assert calls == [False, True]
    assert torch.equal(sel, clean_sel) and torch.equal(labels, clean_labels)
    calls.clear()
    (seeds,) = ms.select_smart_seeds(Xd, 30, first_index=11)
    assert calls == [False, True] and torch.equal(seeds, Xd[clean_sel])


def test_mean_shift_many_seeds():
    from unseenobjectswithmeanshift_amd import synthetic as syn
    X, _ = syn.synth_unit_embeddings(5000, 64, clusters=24, sigma=0.15, seed=2)
    seeds, sel = O.select_smart_seeds(X, 300, 17)
    s2, sel2 = ops().ms_select_seeds(X.to(DEV), 300, 17)
    assert (sel2.cpu() == sel).float().mean() > 0.9
    Zref = O.seed_hill_climbing_ball(X, seeds, 20.0, 3)
    Z = ops().ms_hill_climb(X.to(DEV), seeds.to(DEV), 20.0, 3)
    close(Z, Zref, rtol=1e-4, atol=1e-5)


def _hill_f64(X, Z, kappa, iters):
    X, Z = X.double(), Z.double()
    for _ in range(iters):                                   # MS:90-107 in float64
        Z = torch.nn.functional.normalize(torch.exp(kappa * (Z @ X.t())) @ X, dim=1)
    return Z


@pytest.mark.parametrize("n,S,iters", [(2000, 20, 10), (37, 1, 3), (4111, 50, 4), (5000, 300, 3), (19200, 100, 10), (31, 17, 2)])
def test_mean_shift_hill_climb_split(n, S, iters, lib_option):
    """fp32 results on the bf16 matrix pipe (msm_ms_hill_climb_split): against float64 the error is bounded by 1.5x the
    fp32 MFMA kernel's on the same inputs (ragged n, one seed, seed counts around the block and chunk sizes)."""
    from unseenobjectswithmeanshift_amd import synthetic as syn
    X, _ = syn.synth_unit_embeddings(n, 64, clusters=6, sigma=0.15, seed=n)
    seeds = X[torch.randperm(n, generator=torch.Generator().manual_seed(S))[:S]] if S <= n else X[:S]
    seeds = seeds.contiguous()
    ref = _hill_f64(X, seeds, 20.0, iters)
    z32 = ops().ms_hill_climb(X.to(DEV), seeds.to(DEV), 20.0, iters).cpu().double()
    zsp = ops().ms_hill_climb(X.to(DEV), seeds.to(DEV), 20.0, iters, precision="f32_split").cpu().double()
    e32, esp = (z32 - ref).abs().max().item(), (zsp - ref).abs().max().item()
    print(f"hill climb n={n} S={S}: max |err| vs float64  fp32 MFMA {e32:.2e}  split {esp:.2e}")
    assert esp <= max(1.5 * e32, 2e-7)
    close(zsp.float(), ref.float(), rtol=1e-4, atol=1e-5)
    # the fallback kernel (X split inside the iteration kernel instead of the pre-split planes) multiplies the same terms in the
    # same order: bitwise the same seeds
    lib_option("MS_SPLIT_KERNEL", 1)
    zfb = ops().ms_hill_climb(X.to(DEV), seeds.to(DEV), 20.0, iters, precision="f32_split").cpu().double()
    assert torch.equal(zfb, zsp)
    with pytest.raises(ValueError):
        ops().ms_hill_climb(X.to(DEV), seeds.to(DEV), 20.0, 1, precision="fp8")


@pytest.mark.parametrize("n,S,iters", [(2000, 20, 10), (37, 1, 3), (4111, 50, 4), (5000, 300, 3), (19200, 100, 10), (31, 17, 2), (40000, 161, 5)])
def test_mean_shift_hill_climb_bf16(n, S, iters):
    """msm_ms_hill_climb_bf16 (precision "bf16": one bf16 plane of X, seeds as h + l terms, single bf16 weights): against the float64
    iteration ON THE ROUNDED POINTS (what the kernel is given) the only error left is the rounding of the exp() weights to bf16
    (2^-9 each, averaged over a cluster's points) and fp32 accumulation; against the exact iteration the points' own rounding
    comes on top.  Seed counts around the block sizes (one wave carries up to ten seed blocks: S = 161 -> 11 blocks, 6 per wave)."""
    from unseenobjectswithmeanshift_amd import synthetic as syn
    X, _ = syn.synth_unit_embeddings(n, 64, clusters=6, sigma=0.15, seed=n)
    seeds = X[torch.randperm(n, generator=torch.Generator().manual_seed(S))[:S]] if S <= n else X[:S]
    seeds = seeds.contiguous()
    xb = ops().ms_pack_bf16(X.to(DEV))
    assert xb.dtype == torch.bfloat16 and xb.shape == ((n + 31) // 32 * 32, 64)
    assert torch.equal(xb[:n].cpu(), X.to(torch.bfloat16)) and float(xb[n:].float().abs().sum()) == 0.0
    ref_r = _hill_f64(X.to(torch.bfloat16).float(), seeds, 20.0, iters)
    ref = _hill_f64(X, seeds, 20.0, iters)
    z = ops().ms_hill_climb(X.to(DEV), seeds.to(DEV), 20.0, iters, precision="bf16", xb=xb).cpu().double()
    z2 = ops().ms_hill_climb(X.to(DEV), seeds.to(DEV), 20.0, iters, precision="bf16").cpu().double()          # makes its own copy
    assert torch.equal(z, z2)
    er, ex = (z - ref_r).abs().max().item(), (z - ref).abs().max().item()
    print(f"hill climb bf16 n={n} S={S}: max |err| vs float64 on the rounded points {er:.2e}, on the exact points {ex:.2e}")
    assert er < 2e-3 and ex < 4e-3
    assert float((z.norm(dim=1) - 1).abs().max()) < 1e-5


def test_mean_shift_seeding_bf16_streams_the_copy():
    """msm_ms_select_seeds_bf16 (maps beyond the persistent kernel's 393 216 rows, precision "bf16"): farthest-point seeding on the
    bf16 copy equals the fp32 stepwise kernel run ON THE ROUNDED POINTS index for index (same key, same butterfly; the dot products
    differ only in fp32 summation order: checked on planted clusters where no two candidates tie), and returns rows of the fp32 X."""
    from unseenobjectswithmeanshift_amd import synthetic as syn
    n, S = 400000, 40
    X, _ = syn.synth_unit_embeddings(n, 64, clusters=10, sigma=0.2, seed=5)
    Xd = X.to(DEV)
    xb = ops().ms_pack_bf16(Xd)
    seeds, sel = ops().ms_select_seeds(Xd, S, 123, xb=xb)
    Xr = torch.nn.functional.normalize(X.to(torch.bfloat16).float(), dim=1) * X.to(torch.bfloat16).float().norm(dim=1, keepdim=True)   # = the rounded points
    _, sel_r = ops().ms_select_seeds(Xr.to(DEV).contiguous(), S, 123, stepwise=True)
    assert int(sel[0]) == 123 and sel.unique().numel() == S
    agree = float((sel == sel_r).float().mean())
    print(f"bf16 seeding vs fp32 kernel on the rounded points: {agree:.3f} of the indices equal")
    assert agree >= 0.9                                   # (a near-tie may resolve the other way under another summation order)
    assert torch.equal(seeds.cpu(), X[sel.cpu()])
    # small maps keep the fp32 persistent kernel whatever the precision
    Xs = Xd[:5000].contiguous()
    s1, i1 = ops().ms_select_seeds(Xs, 10, 7, xb=ops().ms_pack_bf16(Xs))
    s2, i2 = ops().ms_select_seeds(Xs, 10, 7)
    assert torch.equal(i1, i2) and torch.equal(s1, s2)


@pytest.mark.parametrize("n", [400000, 1000000, 1000003, 917510])
def test_mean_shift_seeding_bf16_persistent_equals_stepwise(n):
    """The one-launch seeding over the bf16 copy (rows in VGPRs, in LDS, and -- beyond 917 504 rows -- a tail streamed per step)
    selects the SAME indices as one launch per step: every row's arithmetic is shared, only its home differs.  400 000 rows:
    all on chip (tiles past the end re-cover the last rows); 1 000 000: an 82 496-row tail; 1 000 003: a ragged last tile; 917 510: six rows
    beyond the on-chip capacity -- fewer than a tile: the host takes the one-launch-per-step kernel by itself."""
    from unseenobjectswithmeanshift_amd import synthetic as syn
    S = 60
    X, _ = syn.synth_unit_embeddings(n, 64, clusters=12, sigma=0.2, seed=9)
    Xd = X.to(DEV)
    xb = ops().ms_pack_bf16(Xd)
    seeds_p, sel_p = ops().ms_select_seeds(Xd, S, 77, xb=xb)
    seeds_s, sel_s = ops().ms_select_seeds(Xd, S, 77, xb=xb, stepwise=True)
    assert int(sel_p.min()) >= 0, "the persistent kernel gave up on an idle GPU"
    assert torch.equal(sel_p, sel_s) and torch.equal(seeds_p, seeds_s)
    assert sel_p.unique().numel() == S
    # the give-up protocol of the fp32 persistent kernel: indices -1, never a hang or fabricated rows
    _, sel_g = ops().ms_select_seeds(Xd, S, 77, xb=xb, _test_give_up=True)
    if n <= 917504 or n - 917504 >= 16:
        assert bool((sel_g[1:] == -1).all())
    else:                                                   # no persistent launch at this size: nothing to give up
        assert torch.equal(sel_g, sel_s)


# ---------------------------------------------------------------------------------------------
def test_topk_and_postprocess():
    B, Q, h, w, Hh, Ww, T = 2, 100, 30, 40, 120, 160, 20
    logits = rnd(B, Q, 3, seed=1)
    masks = rnd(B, Q, h, w, seed=2, scale=2.0)
    masks[0, 5] = -1.0                                  # an empty mask
    scores, classes, qidx = ops().topk_class_scores(logits.to(DEV), T)
    # the gathering form: same selection, plus the kept rows of a per-query matrix (a column slice of a wider buffer)
    wide = rnd(B, Q, 256, seed=9).to(DEV)
    s2, c2, q2, sel = ops().topk_class_scores(logits.to(DEV), T, gather=wide, gather_cols=68)
    assert torch.equal(s2, scores) and torch.equal(c2, classes) and torch.equal(q2, qidx)
    assert torch.equal(sel, torch.gather(wide[..., :68], 1, qidx.long()[..., None].expand(-1, -1, 68)))
    with pytest.raises(RuntimeError):
        ops().topk_class_scores(logits.to(DEV), T, gather=wide[:, :, ::2], gather_cols=68)
    for b in range(B):
        ref = O.instance_inference(logits[b], masks[b], (Hh, Ww), topk=T)
        sc = torch.softmax(logits[b], -1)[:, :-1].flatten()
        idx = O.canonical_topk(sc, T)
        assert torch.equal(qidx[b].cpu().long(), idx // 2)
        assert torch.equal(classes[b].cpu(), idx % 2)
        close(scores[b], sc[idx], rtol=1e-5, atol=1e-7)
    force = qidx.clone()
    force[0, 0] = 5
    pm, ms, boxes = ops().instance_postprocess(masks.to(DEV), force, (Hh, Ww))
    for b in range(B):
        up = F.interpolate(masks[b][None], size=(Hh, Ww), mode="bilinear", align_corners=False)[0][force[b].cpu().long()]
        binm = (up > 0).float()
        mism = (pm[b].cpu() != binm)
        assert mism.float().mean() < 1e-5
        ref_score = (up.sigmoid().flatten(1) * binm.flatten(1)).sum(1) / (binm.flatten(1).sum(1) + 1e-6)
        close(ms[b], ref_score, rtol=1e-4, atol=1e-5)
        if not mism.any():
            close(boxes[b], O.mask_boxes(up > 0), rtol=0, atol=0)
    assert float(ms[0, 0]) == 0.0 and torch.equal(boxes[0, 0].cpu(), torch.zeros(4))
    # the 4x-specialised strip kernel (register-cached taps) equals the generic one bit for bit
    if Hh == 4 * h and Ww == 4 * w:
        from unseenobjectswithmeanshift_amd._lib import option
        with option("POST_GENERIC", 1):
            pm_g, ms_g, boxes_g = ops().instance_postprocess(masks.to(DEV), force, (Hh, Ww))
        assert torch.equal(pm, pm_g) and torch.equal(boxes, boxes_g)
        close(ms, ms_g.cpu(), rtol=1e-6, atol=1e-7)          # fp32 partial sums are grouped per strip in both, atomics order differs
    # padded frame: upsample to (Hh, Ww), keep the top-left (Hc, Wc) image (PM:275, 354-357); odd widths take the
    # scalar-store path
    for Hc, Wc in ((Hh - 5, Ww - 7), (Hh - 31, Ww), (Hh, Ww - 4)):
        pm, ms, boxes = ops().instance_postprocess(masks.to(DEV), force, (Hc, Wc), padded_size=(Hh, Ww))
        assert pm.shape == (B, T, Hc, Wc)
        for b in range(B):
            up = F.interpolate(masks[b][None], size=(Hh, Ww), mode="bilinear", align_corners=False)[0][force[b].cpu().long()]
            up = up[:, :Hc, :Wc]
            binm = (up > 0).float()
            mism = (pm[b].cpu() != binm)
            assert mism.float().mean() < 1e-5
            ref_score = (up.sigmoid().flatten(1) * binm.flatten(1)).sum(1) / (binm.flatten(1).sum(1) + 1e-6)
            close(ms[b], ref_score, rtol=1e-4, atol=1e-5)
            if not mism.any():
                close(boxes[b], O.mask_boxes(up > 0), rtol=0, atol=0)
    with pytest.raises(RuntimeError):
        ops().instance_postprocess(masks.to(DEV), force, (Hh + 1, Ww), padded_size=(Hh, Ww))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,S,want_next", [(2, 394, True), (1, 50, False), (3, 130, True),
                                           # 1032 / 1031 16-token tiles = 256 four-tile workgroups + 8 / 7 COOPERATIVE
                                           # workgroups (four waves split one tile); ragged last tile
                                           (2, 8249, True), (1, 16496, False)])
def test_encoder_block_fused(B, S, want_next):
    """Fused encoder-layer tail vs the same chain in plain torch fp32 (msdeformattn.py:116-126)."""
    C, DF, PW = 64, 1024, 288
    attn, src, pos = rnd(B, S, C, seed=1), rnd(B, S, C, seed=2), rnd(S, C, seed=3)
    wo, bo = rnd(C, C, seed=4, scale=C ** -0.5), rnd(C, seed=5, scale=0.1)
    w1, b1 = rnd(DF, C, seed=6, scale=C ** -0.5), rnd(DF, seed=7, scale=0.1)
    w2, b2 = rnd(C, DF, seed=8, scale=DF ** -0.5), rnd(C, seed=9, scale=0.1)
    g1, be1, g2, be2 = 1 + 0.1 * rnd(C, seed=10), rnd(C, seed=11, scale=0.1), 1 + 0.1 * rnd(C, seed=12), rnd(C, seed=13, scale=0.1)
    wv, bv = rnd(C, C, seed=14, scale=C ** -0.5), rnd(C, seed=15, scale=0.1)
    wp, bp = rnd(PW, C, seed=16, scale=C ** -0.5), rnd(PW, seed=17)
    x = F.layer_norm(src + F.linear(attn, wo, bo), (C,), g1, be1)
    y = F.layer_norm(x + F.linear(F.relu(F.linear(x, w1, b1)), w2, b2), (C,), g2, be2)
    d = lambda t: t.to(DEV).contiguous()
    stream = ops().pack_encoder_block(d(wo), d(w1), d(w2), d(wv) if want_next else None, d(wp) if want_next else None)
    small = torch.cat([bo, g1, be1, b1, b2, g2, be2, bv, bp]).to(DEV)
    so, vo, po = ops().encoder_block(d(attn), d(src), stream, small, DF, PW, pos=d(pos), tokens_per_image=S, want_next=want_next)
    close(so, y, rtol=1e-4, atol=2e-5)
    if want_next:
        close(vo, F.linear(y, wv, bv), rtol=1e-4, atol=2e-5)
        close(po, F.linear(y + pos, wp, bp), rtol=1e-4, atol=5e-5)
        # head-major value output (B, heads, S, C/heads) is the same data in the layout the gather kernel reads
        so2, vh, po2 = ops().encoder_block(d(attn), d(src), stream, small, DF, PW, pos=d(pos), tokens_per_image=S, value_heads=8)
        assert vh.shape == (B, 8, S, C // 8)
        assert torch.equal(vh.permute(0, 2, 1, 3).reshape(B, S, C), vo) and torch.equal(so2, so) and torch.equal(po2, po)
    else:
        assert vo is None and po is None


@pytest.mark.parametrize("B,S,want_next", [(2, 394, True), (1, 6300, True), (3, 100, False), (8, 6300, True)])
def test_encoder_block_bf16(B, S, want_next):
    """bf16 form of the fused encoder-layer tail (configs 3 / 5): against the chain evaluated in float64 on the bf16-ROUNDED
    operands (weights once; activations where they enter a GEMM) -- what the kernel computes up to fp32 accumulation order --
    and, loosely, against the exact fp32 chain."""
    C, DF, PW = 64, 1024, 288
    attn, src, pos = rnd(B, S, C, seed=1), rnd(B, S, C, seed=2), rnd(S, C, seed=3)
    wo, bo = rnd(C, C, seed=4, scale=C ** -0.5), rnd(C, seed=5, scale=0.1)
    w1, b1 = rnd(DF, C, seed=6, scale=C ** -0.5), rnd(DF, seed=7, scale=0.1)
    w2, b2 = rnd(C, DF, seed=8, scale=DF ** -0.5), rnd(C, seed=9, scale=0.1)
    g1, be1, g2, be2 = 1 + 0.1 * rnd(C, seed=10), rnd(C, seed=11, scale=0.1), 1 + 0.1 * rnd(C, seed=12), rnd(C, seed=13, scale=0.1)
    wv, bv = rnd(C, C, seed=14, scale=C ** -0.5), rnd(C, seed=15, scale=0.1)
    wp, bp = rnd(PW, C, seed=16, scale=C ** -0.5), rnd(PW, seed=17)
    r = lambda t: t.to(torch.bfloat16).double()                       # round to bf16, compute in float64
    lin = lambda t, w, b_: F.linear(r(t.float()), r(w), b_.double())   # linear2: hidden activation and weight single bf16
    lin1 = lambda t, w, b_: F.linear(t.double(), r(w), b_.double())    # linear1: weight bf16, activation hi + lo (exact to 2^-17)
    linx = lambda t, w, b_: F.linear(t.double(), w.double(), b_.double())   # output / value / sampling projections: hi + lo both sides
    x = F.layer_norm(src.double() + linx(attn, wo, bo), (C,), g1.double(), be1.double()).float()
    y = F.layer_norm(x.double() + lin(F.relu(lin1(x, w1, b1)).float(), w2, b2), (C,), g2.double(), be2.double()).float()
    y32 = F.layer_norm(src + F.linear(attn, wo, bo), (C,), g1, be1)
    y32 = F.layer_norm(y32 + F.linear(F.relu(F.linear(y32, w1, b1)), w2, b2), (C,), g2, be2)
    d = lambda t: t.to(DEV).contiguous()
    pack, block = ops().pack_encoder_block_lp, ops().encoder_block_lp      # K = 32 / two-tiles-per-wave kernel (msm_encoder_block_lp_fwd)
    stream = pack(d(wo), d(w1), d(w2), d(wv) if want_next else None, d(wp) if want_next else None)
    small = torch.cat([bo, g1, be1, b1, b2, g2, be2, bv, bp]).to(DEV)
    so, vo, po = block(d(attn), d(src), stream, small, DF, PW, pos=d(pos), tokens_per_image=S, want_next=want_next)
    # an activation next to a bf16 rounding boundary may round the other way (fp32 here, float64 there): one such flip moves
    # the outputs of its token by a few 1e-3; almost all elements agree to fp32 accumulation accuracy
    err = (so.cpu() - y).abs()
    assert float(err.max()) < 1e-2 and float((err > 2e-3).float().mean()) < 1e-4 and float(err.mean()) < 1e-4
    assert float((so.cpu() - y32).abs().max()) < 0.1 and float((so.cpu() - y32).abs().mean()) < 5e-3
    if want_next:
        yk = so.cpu()                                  # the kernel's own layer output feeds its projections
        close(vo, linx(yk, wv, bv).float(), rtol=2e-4, atol=2e-4)        # the three-term products are fp32-class
        close(po, linx(yk + pos, wp, bp).float(), rtol=2e-4, atol=5e-4)
        so2, vh, po2 = block(d(attn), d(src), stream, small, DF, PW, pos=d(pos), tokens_per_image=S, value_heads=8)
        assert vh.shape == (B, 8, S, C // 8)
        assert torch.equal(vh.permute(0, 2, 1, 3).reshape(B, S, C), vo) and torch.equal(so2, so) and torch.equal(po2, po)
    else:
        assert vo is None and po is None


@pytest.mark.parametrize("B,S,want_next", [(2, 394, True), (1, 6300, True), (3, 100, False), (8, 6300, True)])
def test_encoder_block_split_is_fp32_accurate(B, S, want_next):
    """msm_encoder_block_split_fwd: the fp32 encoder-layer tail computed as six bf16 MFMAs per product on exact three-term
    splits of both operands.  It must be an fp32 kernel in everything but the instruction it multiplies with: the same
    tolerances as test_encoder_block_fused against the plain torch fp32 chain, AND its deviation from the float64 chain must
    not exceed the fp32-MFMA kernel's (msm_encoder_block_fwd) by more than 1.5x -- measured side by side, printed."""
    C, DF, PW = 64, 1024, 288
    attn, src, pos = rnd(B, S, C, seed=1), rnd(B, S, C, seed=2), rnd(S, C, seed=3)
    wo, bo = rnd(C, C, seed=4, scale=C ** -0.5), rnd(C, seed=5, scale=0.1)
    w1, b1 = rnd(DF, C, seed=6, scale=C ** -0.5), rnd(DF, seed=7, scale=0.1)
    w2, b2 = rnd(C, DF, seed=8, scale=DF ** -0.5), rnd(C, seed=9, scale=0.1)
    g1, be1, g2, be2 = 1 + 0.1 * rnd(C, seed=10), rnd(C, seed=11, scale=0.1), 1 + 0.1 * rnd(C, seed=12), rnd(C, seed=13, scale=0.1)
    wv, bv = rnd(C, C, seed=14, scale=C ** -0.5), rnd(C, seed=15, scale=0.1)
    wp, bp = rnd(PW, C, seed=16, scale=C ** -0.5), rnd(PW, seed=17)
    D = lambda t: t.double()
    x = F.layer_norm(D(src) + F.linear(D(attn), D(wo), D(bo)), (C,), D(g1), D(be1))
    y = F.layer_norm(x + F.linear(F.relu(F.linear(x, D(w1), D(b1))), D(w2), D(b2)), (C,), D(g2), D(be2))           # float64 chain
    d = lambda t: t.to(DEV).contiguous()
    nxt = (d(wv), d(wp)) if want_next else (None, None)
    small = torch.cat([bo, g1, be1, b1, b2, g2, be2, bv, bp]).to(DEV)
    kw = dict(pos=d(pos), tokens_per_image=S, want_next=want_next)
    so, vo, po = ops().encoder_block_split(d(attn), d(src), ops().pack_encoder_block_split(d(wo), d(w1), d(w2), *nxt), small, DF, PW, **kw)
    s32, v32, p32 = ops().encoder_block(d(attn), d(src), ops().pack_encoder_block(d(wo), d(w1), d(w2), *nxt), small, DF, PW, **kw)
    close(so, y.float(), rtol=1e-4, atol=2e-5)                                     # the fp32 kernel's own tolerances
    e_split, e_mfma = (so.double().cpu() - y).abs(), (s32.double().cpu() - y).abs()
    print(f"encoder block vs float64: split max {float(e_split.max()):.2e} mean {float(e_split.mean()):.2e} | "
          f"fp32 MFMA max {float(e_mfma.max()):.2e} mean {float(e_mfma.mean()):.2e}")
    assert float(e_split.mean()) <= 1.5 * float(e_mfma.mean()) and float(e_split.max()) <= 1.5 * float(e_mfma.max()) + 1e-7
    if want_next:
        yk = so.double().cpu()                              # the kernel's own layer output feeds its projections
        vr, pr = F.linear(yk, D(wv), D(bv)), F.linear(yk + D(pos), D(wp), D(bp))
        close(vo, vr.float(), rtol=1e-4, atol=2e-5)
        close(po, pr.float(), rtol=1e-4, atol=5e-5)
        y32 = s32.double().cpu()
        ev, ev32 = (vo.double().cpu() - vr).abs().mean(), (v32.double().cpu() - F.linear(y32, D(wv), D(bv))).abs().mean()
        ep, ep32 = (po.double().cpu() - pr).abs().mean(), (p32.double().cpu() - F.linear(y32 + D(pos), D(wp), D(bp))).abs().mean()
        print(f"  value_proj mean error split {float(ev):.2e} / fp32 MFMA {float(ev32):.2e}; sampling projection {float(ep):.2e} / {float(ep32):.2e}")
        assert float(ev) <= 1.5 * float(ev32) and float(ep) <= 1.5 * float(ep32)
        so2, vh, po2 = ops().encoder_block_split(d(attn), d(src), ops().pack_encoder_block_split(d(wo), d(w1), d(w2), *nxt), small, DF, PW,
                                                 pos=d(pos), tokens_per_image=S, value_heads=8)
        assert vh.shape == (B, 8, S, C // 8)
        assert torch.equal(vh.permute(0, 2, 1, 3).reshape(B, S, C), vo) and torch.equal(so2, so) and torch.equal(po2, po)
    else:
        assert vo is None and po is None


@pytest.mark.parametrize("B,S,want_next", [(2, 394, True), (1, 6300, True), (3, 100, False), (8, 6300, True), (1, 70000, True)])
def test_encoder_block_hm(B, S, want_next):
    """msm_encoder_block_hm_fwd (bf16 plan, head-major fp16 attn in / value + sampling projection out, one 16-wave workgroup per
    CU): against the chain in float64 on the operands as the kernel rounds them (attn is fp16 data; linear1 weight, linear2
    weight and hidden activation single bf16; everything else hi + lo = exact to 2^-17), and loosely against the exact fp32
    chain.  (1, 70000): more tiles than 256 workgroups x 16 waves -- the grid grows past one workgroup per CU."""
    C, DF, PW = 64, 1024, 288
    attn, src, pos = rnd(B, S, C, seed=1).to(torch.float16).float(), rnd(B, S, C, seed=2), rnd(S, C, seed=3)
    wo, bo = rnd(C, C, seed=4, scale=C ** -0.5), rnd(C, seed=5, scale=0.1)
    w1, b1 = rnd(DF, C, seed=6, scale=C ** -0.5), rnd(DF, seed=7, scale=0.1)
    w2, b2 = rnd(C, DF, seed=8, scale=DF ** -0.5), rnd(C, seed=9, scale=0.1)
    g1, be1, g2, be2 = 1 + 0.1 * rnd(C, seed=10), rnd(C, seed=11, scale=0.1), 1 + 0.1 * rnd(C, seed=12), rnd(C, seed=13, scale=0.1)
    wv, bv = rnd(C, C, seed=14, scale=C ** -0.5), rnd(C, seed=15, scale=0.1)
    wp, bp = rnd(PW, C, seed=16, scale=C ** -0.5), rnd(PW, seed=17)
    r = lambda t: t.to(torch.bfloat16).double()
    lin = lambda t, w, b_: F.linear(r(t.float()), r(w), b_.double())
    lin1 = lambda t, w, b_: F.linear(t.double(), r(w), b_.double())
    linx = lambda t, w, b_: F.linear(t.double(), w.double(), b_.double())
    x = F.layer_norm(src.double() + linx(attn, wo, bo), (C,), g1.double(), be1.double()).float()
    y = F.layer_norm(x.double() + lin(F.relu(lin1(x, w1, b1)).float(), w2, b2), (C,), g2.double(), be2.double()).float()
    y32 = F.layer_norm(src + F.linear(attn, wo, bo), (C,), g1, be1)
    y32 = F.layer_norm(y32 + F.linear(F.relu(F.linear(y32, w1, b1)), w2, b2), (C,), g2, be2)
    d = lambda t: t.to(DEV).contiguous()
    nxt = (d(wv), d(wp)) if want_next else (None, None)
    stream = ops().pack_encoder_block_hm(d(wo), d(w1), d(w2), *nxt)
    small = ops().pack_encoder_block_hm_small(*[d(t) for t in (bo, g1, be1, b1, b2, g2, be2)], *((d(bv), d(bp)) if want_next else (None, None)))
    attn_hm = d(attn.view(B, S, 8, 8).permute(0, 2, 1, 3).to(torch.float16))
    so, vh, ph = ops().encoder_block_hm(attn_hm, d(src), stream, small, DF, pos=d(pos), want_next=want_next)
    err = (so.cpu() - y).abs()
    assert float(err.max()) < 1e-2 and float((err > 2e-3).float().mean()) < 1e-4 and float(err.mean()) < 1e-4
    assert float((so.cpu() - y32).abs().max()) < 0.1 and float((so.cpu() - y32).abs().mean()) < 5e-3
    if want_next:
        assert vh.shape == (B, 8, S, 8) and vh.dtype == torch.float16 and ph.shape == (B, 8, S, 30) and ph.dtype == torch.float32
        yk = so.cpu()                                          # the kernel's own layer output feeds its projections
        got = vh.float().cpu().permute(0, 2, 1, 3).reshape(B, S, C)
        close(got, linx(yk, wv, bv).float(), rtol=2 ** -10, atol=2e-4)         # one fp16 rounding of an fp32-class result
        # the sampling records (round 5): 24 offsets in fp32 -- an fp32-class result (hi + lo operands: 2^-17) --, 12 logits in fp16
        pexact = linx(yk + pos, wp, bp).float()
        pgot = ops().proj_records_to_columns(ph).cpu()
        close(pgot[..., :192], pexact[..., :192], rtol=1e-4, atol=1e-4)        # (three-term bf16 products: 2^-17 per term on O(1 .. 10) sums)
        close(pgot[..., 192:], pexact[..., 192:], rtol=2 ** -10, atol=2e-4)    # (a value on a rounding boundary may round the other way)
        assert torch.equal(ops().proj_records_to_columns(ops().proj_to_head_major_records(d(pexact))).cpu()[..., :192], pexact[..., :192])
    else:
        assert vh is None and ph is None
    # precision "f16" (ffn_f16): W1 / W2 / x / the hidden activation as IEEE halves, one term each -- against the float64 chain on
    # the operands as THAT form rounds them, and at least three times closer to the exact fp32 chain than the bf16 form
    h16 = lambda t: t.to(torch.float16).double()
    hid = F.relu(F.linear(h16(x), h16(w1), b1.double())).float()
    yh = F.layer_norm(x.double() + F.linear(h16(hid), h16(w2), b2.double()), (C,), g2.double(), be2.double()).float()
    stream_h = ops().pack_encoder_block_hm(d(wo), d(w1), d(w2), *nxt, ffn_f16=True)
    assert stream_h.shape == stream.shape and not torch.equal(stream_h, stream)
    soh, _, _ = ops().encoder_block_hm(attn_hm, d(src), stream_h, small, DF, pos=d(pos), want_next=want_next, ffn_f16=True)
    errh = (soh.cpu() - yh).abs()
    assert float(errh.max()) < 2e-3 and float(errh.mean()) < 3e-5
    e_h, e_b = float((soh.cpu() - y32).abs().mean()), float((so.cpu() - y32).abs().mean())
    print(f"encoder block hm B={B} S={S}: mean |err| against the exact fp32 chain  bf16 FFN {e_b:.2e}  fp16 FFN {e_h:.2e}")
    assert 3 * e_h <= e_b
    # round 6: the default is the half-size form (enc_block_hm2_kernel: eight waves, 16-KiB stages of the SAME stream, two workgroups per CU);
    # the one-workgroup form (option ENC_NO_COOP = 2) returns the same bits -- a tile's arithmetic does not depend on the staging
    from unseenobjectswithmeanshift_amd._lib import option
    with option("ENC_NO_COOP", 2):
        so1, vh1, ph1 = ops().encoder_block_hm(attn_hm, d(src), stream, small, DF, pos=d(pos), want_next=want_next)
        soh1, _, _ = ops().encoder_block_hm(attn_hm, d(src), stream_h, small, DF, pos=d(pos), want_next=want_next, ffn_f16=True)
    assert torch.equal(so1, so) and torch.equal(soh1, soh)
    if want_next:
        assert torch.equal(vh1, vh) and torch.equal(ph1.view(torch.int32), ph.view(torch.int32))
    with pytest.raises(RuntimeError):
        ops().encoder_block_hm(attn_hm.float(), d(src), stream, small, DF, pos=d(pos), want_next=want_next)
    with pytest.raises(RuntimeError):
        ops().encoder_block_hm(attn_hm, d(src), stream, small, DF, pos=d(pos), want_next=not want_next)      # stream / plan mismatch


@pytest.mark.parametrize("B,shp", [(2, [(6, 8), (12, 16), (24, 32)]), (8, [(15, 20), (30, 40), (60, 80)]), (1, [(2, 3), (4, 6), (8, 12)])])
def test_msda_encoder_lp(B, shp):
    """The bf16 plan's gathers (fp16 value taps, fp16 result; sampling projection read as head-major 120-byte records, or computed in
    the kernel from src + pos) against the fp32 pair they replace -- F.linear for [sampling_offsets | attention_weights](src + pos),
    msm_msdeform_attn_enc_hm_fwd on the same (bf16-valued) value."""
    M, D, L, P = 8, 8, 3, 4
    S = sum(h * w for h, w in shp)
    shapes = torch.tensor(shp, dtype=torch.int64)
    start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    value = rnd(B, M, S, D, seed=1).to(torch.float16)
    src, pos = rnd(B, S, 64, seed=2), rnd(S, 64, seed=3)
    wp, bp = rnd(288, 64, seed=4, scale=0.05), rnd(288, seed=5)
    wp[:192] *= 4.0                                              # offsets of a few pixels
    d = lambda t: t.to(DEV).contiguous()
    proj = F.linear((src + pos).double(), wp.double(), bp.double()).float()
    # stored projection: the reference sees the fp32 offsets / fp16-rounded logits the kernel reads -> only the result's rounding is left
    proj_hm = ops().proj_to_head_major_records(d(proj))
    assert proj_hm.shape == (B, M, S, 30) and proj_hm.dtype == torch.float32
    proj_r = ops().proj_records_to_columns(proj_hm)              # back to the reference's column order
    assert torch.equal(proj_r[..., :192].cpu(), proj[..., :192])
    ref = ops().ms_deform_attn_encoder(d(value.float()), d(shapes), d(start), proj_r, M, P)      # (B, S, 64)
    ref_hm = ref.view(B, S, M, D).permute(0, 2, 1, 3).cpu()
    got = ops().ms_deform_attn_encoder_lp(d(value), d(shapes), d(start), proj_hm, P)
    assert got.shape == (B, M, S, D) and got.dtype == torch.float16
    close(got.float().cpu(), ref_hm, rtol=2 ** -10, atol=2e-5)
    # projection in the kernel: hi + lo split of the projection (2^-17) and the rounding of the result
    wpack, bpack = ops().pack_msda_proj_lp(d(wp), d(bp))
    got = ops().ms_deform_attn_encoder_lp_fused(d(value), d(shapes), d(start), d(src), d(pos), wpack, bpack, P)
    ref = ops().ms_deform_attn_encoder(d(value.float()), d(shapes), d(start), d(proj), M, P)
    ref_hm = ref.view(B, S, M, D).permute(0, 2, 1, 3).cpu()
    err = (got.float().cpu() - ref_hm).abs()
    assert float(err.max()) < 5e-3 and float(err.mean()) < 3e-4, (float(err.max()), float(err.mean()))
    # fp32 -> fp16 conversion entry point (clamped to the half range)
    t = rnd(3, 8, 50, 8, seed=9)
    t[0, 0, 0, 0], t[0, 0, 0, 1] = 1e6, -1e6
    assert torch.equal(ops().to_f16(d(t)).cpu(), t.clamp(-65504, 65504).to(torch.float16))


def test_pixel_decoder_fused_equals_unfused():
    from unseenobjectswithmeanshift_amd import synthetic as syn
    from unseenobjectswithmeanshift_amd.meta_arch import build_resnet50_head
    head = build_resnet50_head()
    head.pixel_decoder.load_state_dict(syn.synth_state_dict(syn.pixel_decoder_param_shapes()), strict=True)
    pd = head.pixel_decoder.to(DEV).eval()
    feats = {k: v.to(DEV) for k, v in syn.synth_backbone_features(2, 64, 96, seed=3).items()}
    pd.fused_encoder = True
    a = pd.forward_features(feats)
    pd.fused_encoder = False
    b = pd.forward_features(feats)
    torch.testing.assert_close(a[0], b[0], rtol=1e-4, atol=5e-5)
    for x, y in zip(a[2], b[2]):
        torch.testing.assert_close(x, y, rtol=1e-4, atol=5e-5)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Cin,H,W", [(2, 2048, 15, 20), (3, 1024, 30, 40), (8, 512, 60, 80), (1, 256, 10, 6), (2, 128, 7, 4),
                                        (2, 256, 100, 164), (1, 384, 128, 260)])      # the last two: the shallow-K kernel, ragged last tile
def test_conv1x1_in_vs_fp64(B, Cin, H, W):
    """msm_conv1x1_in_f32 (every tile width the host picks, ragged last tiles) against an fp64 1x1 convolution and the
    fp64 moments of its own output; writing into a slice of a larger token buffer; moment accumulation."""
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(64, Cin, seed=2, scale=Cin ** -0.5), rnd(64, seed=3)
    ref = torch.einsum("bchw,oc->bhwo", x.double(), w.double()).reshape(B, H * W, 64) + b.double()
    wp = ops().pack_conv_in_weight(w.to(DEV))
    # the packed layout is the one include/msm_hip.h documents
    k, o = torch.meshgrid(torch.arange(Cin), torch.arange(64), indexing="ij")
    idx = (((k // 8) * 4 + o // 16) * 64 + ((k % 8) // 2) * 16 + o % 16) * 2 + k % 2
    assert torch.equal(wp.cpu()[idx], w.t())
    from unseenobjectswithmeanshift_amd._lib import set_option
    for nt in (1, 2, 4, -1):
        set_option("CONVIN_NT", nt)
        out, st = ops().conv1x1_in(x.to(DEV), wp, b.to(DEV))
        closed(out, ref, rtol=2e-5, atol=2e-5)
        mom = torch.stack([out.double().sum(1), (out.double() ** 2).sum(1)], -1).cpu()
        torch.testing.assert_close(st.cpu(), mom, rtol=1e-5, atol=1e-4)     # fp32 partial sums per 16..64-pixel tile
    # slice of a concatenated buffer, no bias, moments accumulated on top of what the caller put there
    buf = torch.full((B, H * W + 24, 64), 7.0, device=DEV)
    st0 = torch.ones(B, 64, 2, device=DEV, dtype=torch.float64)
    out, st = ops().conv1x1_in(x.to(DEV), wp, None, out=buf[:, 8:8 + H * W], stats=st0, stats_cleared=True)
    closed(out, ref - b.double(), rtol=2e-5, atol=2e-5)
    assert float(buf[:, :8].min()) == 7.0 == float(buf[:, 8 + H * W:].max()) and st.data_ptr() == st0.data_ptr()
    mom = torch.stack([out.double().sum(1), (out.double() ** 2).sum(1)], -1).cpu() + 1.0
    torch.testing.assert_close(st.cpu(), mom, rtol=1e-5, atol=1e-4)
    # run to run identical (fixed-order reduction over the K slices)
    again, _ = ops().conv1x1_in(x.to(DEV), wp, None)
    assert torch.equal(again, out)
    with pytest.raises(RuntimeError):
        ops().conv1x1_in(torch.zeros(1, 96, 4, 4, device=DEV), torch.zeros(64 * 96, device=DEV))


@pytest.mark.parametrize("B,Cin,H,W", [(2, 2048, 15, 20), (3, 1024, 30, 40), (8, 512, 60, 80), (1, 256, 10, 6), (2, 256, 100, 164),
                                        (1, 512, 128, 65)])          # the last two: one tile per wave over the full K, ragged last tile
def test_conv1x1_in_lp_vs_fp64(B, Cin, H, W):
    """msm_conv1x1_in_lp (the bf16 plan's input projections: hi + lo bf16 operands, three K = 32 MFMAs per product, fp32 results)
    against the fp64 convolution to the FP32 kernel's tolerance, its moments against the fp64 moments of its own output, the packed
    layout against the header's formula, a slice of a larger buffer, and the multi-level launch against single launches."""
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(64, Cin, seed=2, scale=Cin ** -0.5), rnd(64, seed=3)
    ref = torch.einsum("bchw,oc->bhwo", x.double(), w.double()).reshape(B, H * W, 64) + b.double()
    wp = ops().pack_conv_in_weight_lp(w.to(DEV))
    hi = w.to(torch.bfloat16)
    planes = torch.stack([hi, (w - hi.float()).to(torch.bfloat16)])
    k, o = torch.meshgrid(torch.arange(Cin), torch.arange(64), indexing="ij")
    for pl in range(2):
        idx = ((((k // 32) * 4 + o // 16) * 2 + pl) * 64 + ((k % 32) // 8) * 16 + o % 16) * 8 + k % 8
        assert torch.equal(wp.cpu()[idx], planes[pl].t())
    out, st = ops().conv1x1_in(x.to(DEV), wp, b.to(DEV), lp=True)
    closed(out, ref, rtol=6e-5, atol=6e-5)                 # (operands carry 16 mantissa bits: 3x the fp32 kernel's bound)
    e_lp = float((out.double().cpu() - ref).abs().mean())
    out32, _ = ops().conv1x1_in(x.to(DEV), ops().pack_conv_in_weight(w.to(DEV)), b.to(DEV))
    e_32 = float((out32.double().cpu() - ref).abs().mean())
    print(f"mean |error| against float64: hi+lo bf16 {e_lp:.3e}, fp32 MFMA {e_32:.3e}")
    assert e_lp <= 1e-5              # operands carry 16 mantissa bits whatever K: three orders of magnitude under one bf16 rounding (2e-3)
    mom = torch.stack([out.double().sum(1), (out.double() ** 2).sum(1)], -1).cpu()
    torch.testing.assert_close(st.cpu(), mom, rtol=1e-5, atol=1e-4)
    buf = torch.full((B, H * W + 24, 64), 7.0, device=DEV)
    st0 = torch.ones(B, 64, 2, device=DEV, dtype=torch.float64)
    out, st = ops().conv1x1_in(x.to(DEV), wp, None, out=buf[:, 8:8 + H * W], stats=st0, stats_cleared=True, lp=True)
    closed(out, ref - b.double(), rtol=6e-5, atol=6e-5)
    assert float(buf[:, :8].min()) == 7.0 == float(buf[:, 8 + H * W:].max()) and st.data_ptr() == st0.data_ptr()
    torch.testing.assert_close(st.cpu(), torch.stack([out.double().sum(1), (out.double() ** 2).sum(1)], -1).cpu() + 1.0, rtol=1e-5, atol=1e-4)
    again, _ = ops().conv1x1_in(x.to(DEV), wp, None, lp=True)
    assert torch.equal(again, out)                        # fixed-order reduction over the K slices
    with pytest.raises(RuntimeError):
        ops().conv1x1_in(torch.zeros(1, 128, 4, 4, device=DEV), torch.zeros(128 * 128, device=DEV, dtype=torch.bfloat16), lp=True)


@pytest.mark.parametrize("B,levels", [(8, ((2048, 15, 20), (1024, 30, 40), (512, 60, 80))), (3, ((2048, 4, 6), (1024, 8, 12), (512, 16, 24))),
                                      (2, ((512, 17, 20),)), (5, ((1024, 7, 12), (256, 14, 24))), (1, ((256, 2, 2), (256, 9, 8)))])
def test_conv1x1_in_multi_wide_vs_fp64(B, levels):
    """msm_conv1x1_in_multi_wide (round 6: the deep input projections of the 16-bit plans with the packed weight broadcast through LDS --
    eight-wave workgroups over adjacent 64-pixel tiles x K slices, LDS-DMA ring, counted waits): against the fp64 convolution to the lp
    form's tolerance (the same hi + lo operands), moments of its own output, ragged tiles (300 pixels), one / two / four K slices, biases
    present and absent, a token-range view of a larger buffer, and run twice (fixed-order sums: the same bits)."""
    xs = [rnd(B, c, h, w, seed=20 + i) for i, (c, h, w) in enumerate(levels)]
    w32 = [rnd(64, x.shape[1], seed=30 + i, scale=x.shape[1] ** -0.5) for i, x in enumerate(xs)]
    bs = [rnd(64, seed=40 + i) if i != 1 else None for i in range(len(xs))]
    ws = [ops().pack_conv_in_weight_lp(w.to(DEV)) for w in w32]
    S = sum(x.shape[2] * x.shape[3] for x in xs)
    buf = torch.full((B, S + 24, 64), 7.0, device=DEV)
    out = buf[:, 8:8 + S]
    st = torch.ones(len(xs), B, 64, 2, device=DEV, dtype=torch.float64)
    xd, bd = [x.to(DEV) for x in xs], [None if b is None else b.to(DEV) for b in bs]
    ops().conv1x1_in_multi(xd, ws, bd, out, st, stats_cleared=True, lp="wide")
    assert float(buf[:, :8].min()) == 7.0 == float(buf[:, 8 + S:].max())
    o = 0
    for l, x in enumerate(xs):
        hw = x.shape[2] * x.shape[3]
        ref = torch.einsum("bchw,oc->bhwo", x.double(), w32[l].double()).reshape(B, hw, 64) + (0 if bs[l] is None else bs[l].double())
        got = out[:, o:o + hw]
        closed(got, ref, rtol=6e-5, atol=6e-5)
        assert float((got.double().cpu() - ref).abs().mean()) <= 1e-5
        mom = torch.stack([got.double().sum(1), (got.double() ** 2).sum(1)], -1).cpu()
        torch.testing.assert_close(st[l].cpu(), mom + 1.0, rtol=1e-5, atol=1e-4)
        o += hw
    first = out.clone()
    st2 = torch.zeros_like(st)
    ops().conv1x1_in_multi(xd, ws, bd, out, st2, stats_cleared=True, lp="wide")
    assert torch.equal(out, first)


def test_conv1x1_in_multi_lp_equals_single_launches():
    B = 3
    xs = [rnd(B, c, h, w, seed=20 + i).to(DEV) for i, (c, h, w) in enumerate(((2048, 4, 6), (1024, 8, 12), (512, 16, 24)))]
    ws = [ops().pack_conv_in_weight_lp(rnd(64, x.shape[1], seed=30 + i, scale=x.shape[1] ** -0.5).to(DEV)) for i, x in enumerate(xs)]
    bs = [rnd(64, seed=40).to(DEV), None, rnd(64, seed=42).to(DEV)]
    S = sum(x.shape[2] * x.shape[3] for x in xs)
    out = torch.empty(B, S, 64, device=DEV)
    st = torch.zeros(3, B, 64, 2, device=DEV, dtype=torch.float64)
    ops().conv1x1_in_multi(xs, ws, bs, out, st, stats_cleared=True, lp=True)
    o = 0
    for l, x in enumerate(xs):
        hw = x.shape[2] * x.shape[3]
        ref, rst = ops().conv1x1_in(x, ws[l], bs[l], lp=True)
        assert torch.equal(out[:, o:o + hw], ref)
        torch.testing.assert_close(st[l], rst, rtol=1e-12, atol=1e-9)
        o += hw


def test_encoder_prologue_vs_fp64():
    """msm_encoder_prologue_fwd: GroupNorm of three concatenated levels from the conv moments, value projection
    (token- and head-major) and sampling projections of src + pos, against fp64 torch ops."""
    B, shapes, pw = 3, ((3, 4), (6, 8), (12, 16)), 288
    S = sum(h * w for h, w in shapes)
    raw = rnd(B, S, 64, seed=4, scale=2.0) + 0.5
    gam, bet = rnd(3, 64, seed=5) * 0.2 + 1.0, rnd(3, 64, seed=6) * 0.3
    wv, bv, wp, bp = rnd(64, 64, seed=7, scale=0.2), rnd(64, seed=8), rnd(pw, 64, seed=9, scale=0.2), rnd(pw, seed=10)
    pos = rnd(S, 64, seed=11)
    bounds, parts, stats = [0], [], []
    for l, (h, w) in enumerate(shapes):
        seg = raw[:, bounds[-1]:bounds[-1] + h * w].double()
        bounds.append(bounds[-1] + h * w)
        stats.append(torch.stack([seg.sum(1), (seg ** 2).sum(1)], -1))
        y = F.group_norm(seg.transpose(1, 2), 32, gam[l].double(), bet[l].double(), 1e-5).transpose(1, 2)
        parts.append(y)
    src_ref = torch.cat(parts, 1)
    val_ref = src_ref @ wv.double().t() + bv.double()
    proj_ref = (src_ref + pos.double()) @ wp.double().t() + bp.double()
    o = ops()
    stream = o.pack_encoder_prologue(wv.to(DEV), wp.to(DEV))
    small = torch.cat([bv, bp]).to(DEV)
    gnp = torch.stack([gam, bet], 1).contiguous().to(DEV)
    st = torch.stack(stats).to(DEV)
    for heads in (0, 8):
        src, value, proj = o.encoder_prologue(raw.clone().to(DEV), st, gnp, bounds, stream, small, pos.to(DEV), pw, value_heads=heads)
        closed(src, src_ref, rtol=2e-5, atol=2e-5)
        closed(proj, proj_ref, rtol=5e-5, atol=5e-5)
        if heads:
            value = value.permute(0, 2, 1, 3).reshape(B, S, 64)
        closed(value, val_ref, rtol=5e-5, atol=5e-5)
    # the bf16 plan's outputs: the same fp32 results rounded once, in the head-major layouts of csrc/enc_lp.hip
    src, value, proj = o.encoder_prologue(raw.clone().to(DEV), st, gnp, bounds, stream, small, pos.to(DEV), pw, value_heads=8)
    s2, v2, p2 = o.encoder_prologue(raw.clone().to(DEV), st, gnp, bounds, stream, small, pos.to(DEV), pw, value_heads=8, bf16_hm=True)
    assert torch.equal(s2, src) and torch.equal(v2, value.to(torch.float16)) and torch.equal(p2, o.proj_to_head_major_records(proj))
    # ... and the bf16 plan's own prologue (msm_encoder_prologue_hm_fwd: the two projections on the bf16 matrix pipe with hi + lo
    # operands): the same src bit for bit, value / record equal to the fp64 reference to fp16 storage precision
    blocks, small_hm = o.pack_encoder_prologue_hm(wv.to(DEV), wp.to(DEV), bv.to(DEV), bp.to(DEV))
    s3, v3, p3 = o.encoder_prologue_hm(raw.clone().to(DEV), st, gnp, bounds, blocks, small_hm, pos.to(DEV))
    assert torch.equal(s3, src) and v3.dtype == torch.float16 and p3.dtype == torch.float32 and p3.shape == (B, 8, S, 30)
    val_hm = val_ref.view(B, S, 8, 8).permute(0, 2, 1, 3)
    closed(v3.float(), val_hm, rtol=1.5e-3, atol=1.5e-3)              # (an fp16 rounding of O(1) values: 2^-11 relative)
    c3, c2 = o.proj_records_to_columns(p3), o.proj_records_to_columns(p2)
    closed(c3[..., :192], proj_ref[..., :192], rtol=5e-5, atol=5e-5)  # fp32 offsets from hi + lo operands: an fp32-class result
    closed(c3[..., 192:], proj_ref[..., 192:], rtol=1.5e-3, atol=1.5e-3)
    # against the fp32-MFMA prologue's outputs: value / logits at most one fp16 step apart, offsets to fp32 rounding
    assert float((v3.float() - v2.float()).abs().max()) <= 4e-3 and float((c3[..., 192:] - c2[..., 192:]).abs().max()) <= 4e-3
    assert float((c3[..., :192] - c2[..., :192]).abs().max()) <= 1e-4
    with pytest.raises(RuntimeError):
        o.encoder_prologue(raw.to(DEV), st, gnp, [0, 12, 60, S + 1], stream, small, pos.to(DEV), pw)


def test_conv1x1_in_multi_equals_single_launches():
    """msm_conv1x1_in_multi_f32 (all levels in one launch) is bit-identical to one msm_conv1x1_in_f32 per level."""
    B = 3
    xs = [rnd(B, c, h, w, seed=20 + i).to(DEV) for i, (c, h, w) in enumerate(((2048, 4, 6), (1024, 8, 12), (512, 16, 24)))]
    ws = [ops().pack_conv_in_weight(rnd(64, x.shape[1], seed=30 + i, scale=x.shape[1] ** -0.5).to(DEV)) for i, x in enumerate(xs)]
    bs = [rnd(64, seed=40).to(DEV), None, rnd(64, seed=42).to(DEV)]
    S = sum(x.shape[2] * x.shape[3] for x in xs)
    out = torch.empty(B, S, 64, device=DEV)
    st = torch.zeros(3, B, 64, 2, device=DEV, dtype=torch.float64)
    ops().conv1x1_in_multi(xs, ws, bs, out, st, stats_cleared=True)
    o = 0
    for l, x in enumerate(xs):
        hw = x.shape[2] * x.shape[3]
        ref, rst = ops().conv1x1_in(x, ws[l], bs[l])
        assert torch.equal(out[:, o:o + hw], ref)
        torch.testing.assert_close(st[l], rst, rtol=1e-12, atol=1e-9)
        o += hw


@pytest.mark.parametrize("B,H,W", [(2, 12, 16), (1, 5, 37), (3, 30, 40)])
def test_conv3x3_c64_vs_fp64(B, H, W):
    """msm_conv3x3_c64_f32 (weight held in LDS) against an fp64 conv2d with zero padding, ragged widths included, and the
    fp64 moments of its own output; equal to the implicit-GEMM path up to summation order."""
    x, w = rnd(B, H * W, 64, seed=1), rnd(64, 64, 3, 3, seed=2, scale=0.06)
    ref = F.conv2d(x.double().view(B, H, W, 64).permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, 64)
    w3 = w.permute(0, 2, 3, 1).reshape(64, 576).contiguous().to(DEV)
    out, st = ops().conv3x3_c64(x.to(DEV), w3, H, W)
    closed(out, ref, rtol=2e-5, atol=2e-5)
    mom = torch.stack([out.double().sum(1), (out.double() ** 2).sum(1)], -1).cpu()
    torch.testing.assert_close(st.cpu(), mom, rtol=1e-5, atol=1e-4)
    close(out, ops().conv3x3_tokens(x.to(DEV), w3, H, W).cpu(), rtol=2e-5, atol=2e-5)
    st0 = torch.ones(B, 64, 2, device=DEV, dtype=torch.float64)
    out2, st2 = ops().conv3x3_c64(x.to(DEV), w3, H, W, stats=st0, stats_cleared=True)
    assert torch.equal(out2, out) and st2.data_ptr() == st0.data_ptr()
    torch.testing.assert_close(st2.cpu(), mom + 1.0, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,H,W", [(2, 12, 16), (1, 5, 37), (3, 30, 40), (8, 120, 160), (40, 7, 33)])
def test_conv3x3_c64_bf16_mode(B, H, W):
    """msm_conv3x3_c64_bf16 (low-precision mode: weight rounded to bf16, activations as hi + lo operands, fp32 accumulation)
    against the fp64 convolution WITH THE SAME ROUNDED WEIGHT to the fp32 kernel's tolerance -- the activations' hi + lo pair
    carries 16 mantissa bits --, against the unrounded fp64 convolution to bf16's, and the moments of its own output."""
    x, w = rnd(B, H * W, 64, seed=1), rnd(64, 64, 3, 3, seed=2, scale=0.06)
    xi = x.double().view(B, H, W, 64).permute(0, 3, 1, 2)
    ref = F.conv2d(xi, w.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, 64)
    ref_r = F.conv2d(xi, w.bfloat16().double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, 64)
    w3 = w.permute(0, 2, 3, 1).reshape(64, 576).contiguous().to(DEV)
    out, st = ops().conv3x3_c64(x.to(DEV), w3, H, W, bf16=True)
    closed(out, ref_r, rtol=1e-4, atol=1e-4)
    err = float((out.cpu().double() - ref).abs().max())
    assert err < 2e-2 * float(ref.abs().max()), err
    mom = torch.stack([out.double().sum(1), (out.double() ** 2).sum(1)], -1).cpu()
    torch.testing.assert_close(st.cpu(), mom, rtol=1e-5, atol=1e-4)
    # (round 6: maps at least 32 pixels wide take the one-row-per-unit kernel -- rows requested together, split once, dx taps as lane
    # shifts; the per-tap kernel it replaces, MSM_OPT_CONV3_WIDE = 1, returns the same bits)
    from unseenobjectswithmeanshift_amd import _lib
    with _lib.option("CONV3_WIDE", 1):
        out_tap, st_tap = ops().conv3x3_c64(x.to(DEV), w3, H, W, bf16=True)
    assert torch.equal(out_tap, out)
    torch.testing.assert_close(st_tap.cpu(), st.cpu(), rtol=1e-5, atol=1e-4)            # (fp32 partial sums grouped by unit)
    # precision "f16" (msm_conv3x3_c64_f16): weight and activations one IEEE-half term each -- against the fp64 convolution of the
    # fp16-ROUNDED operands to the fp32 kernel's tolerance, and against the exact one at least four times closer than the bf16 form
    h16 = lambda t: t.to(torch.float16).double()
    ref_h = F.conv2d(h16(x).view(B, H, W, 64).permute(0, 3, 1, 2), h16(w), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, 64)
    outh, sth = ops().conv3x3_c64(x.to(DEV), w3, H, W, bf16="f16")
    closed(outh, ref_h, rtol=1e-4, atol=1e-4)
    e_h, e_b = float((outh.cpu().double() - ref).abs().mean()), float((out.cpu().double() - ref).abs().mean())
    print(f"conv3x3 {B}x{H}x{W}: mean |err| vs exact fp64  bf16 form {e_b:.2e}  f16 form {e_h:.2e}")
    assert 4 * e_h <= e_b
    momh = torch.stack([outh.double().sum(1), (outh.double() ** 2).sum(1)], -1).cpu()
    torch.testing.assert_close(sth.cpu(), momh, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,H,W", [(2, 12, 16), (1, 5, 37), (3, 30, 40), (8, 120, 160)])
def test_conv3x3_c64_split_form(B, H, W):
    """f32_split: GroupNorm written as three bf16 planes (exact: h + m + l == the fp32 result bit for bit) and the 3x3
    convolution as six bf16 MFMAs per product (msm_groupnorm_apply_split + msm_conv3x3_c64_split): against the fp64
    convolution the error is bounded by 1.5x the fp32 MFMA kernel's, at ragged widths and at the headline size."""
    x, w = rnd(B, H * W, 64, seed=1), rnd(64, 64, 3, 3, seed=2, scale=0.06)
    g, be = 1 + 0.1 * rnd(64, seed=3), rnd(64, seed=4)
    up = rnd(B, (H // 2) * (W // 2), 64, seed=5) if H % 2 == 0 and W % 2 == 0 else None
    kw = dict(up=up.to(DEV), up_hw=(H // 2, W // 2)) if up is not None else {}
    y32 = ops().groupnorm_tokens(x.to(DEV), g.to(DEV), be.to(DEV), H, W, **kw)
    planes = ops().groupnorm_tokens(x.to(DEV), g.to(DEV), be.to(DEV), H, W, split_planes=True, **kw)
    assert planes.shape == (3, B, H * W, 64) and planes.dtype == torch.bfloat16
    assert torch.equal((planes[0].float() + planes[1].float()) + planes[2].float(), y32)
    ref = F.conv2d(y32.cpu().double().view(B, H, W, 64).permute(0, 3, 1, 2), w.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, 64)
    w3 = w.permute(0, 2, 3, 1).reshape(64, 576).contiguous().to(DEV)
    o32, st32 = ops().conv3x3_c64(y32, w3, H, W)
    osp, stsp = ops().conv3x3_c64(planes, w3, H, W, split=True)
    e32, esp = float((o32.cpu().double() - ref).abs().max()), float((osp.cpu().double() - ref).abs().max())
    print(f"conv3x3 {B}x{H}x{W}: max |err| vs float64  fp32 MFMA {e32:.2e}  split {esp:.2e}")
    assert esp <= 1.5 * e32 + 1e-7
    mom = torch.stack([osp.double().sum(1), (osp.double() ** 2).sum(1)], -1).cpu()
    torch.testing.assert_close(stsp.cpu(), mom, rtol=1e-5, atol=1e-4)
    st0 = torch.ones(B, 64, 2, device=DEV, dtype=torch.float64)
    o2, st2 = ops().conv3x3_c64(planes, w3, H, W, split=True, stats=st0, stats_cleared=True)
    assert torch.equal(o2, osp)
    torch.testing.assert_close(st2.cpu(), mom + 1.0, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,H,W", [(2, 12, 16), (1, 5, 37), (3, 30, 40), (8, 120, 160), (20, 56, 56), (2, 240, 320), (1, 1, 4), (40, 7, 33)])
def test_conv3x3_c64_half_map_form(B, H, W):
    """The "f16" plan's FPN level (round 6): msm_groupnorm_apply_f16 writes the clamped IEEE halves msm_conv3x3_c64_f16 rounds its input
    to -- bit for bit --, and msm_conv3x3_c64_f16h (a wave = up to three output rows of a 32-pixel strip, every load of the unit in flight
    at once, dx = -1 / +1 operands as lane shifts) returns msm_conv3x3_c64_f16's output bits on them: ragged widths, one-row maps, more
    images than workgroup slots, the headline size; moments of its own output; accumulation into a caller's zeroed moments."""
    x, w = rnd(B, H * W, 64, seed=1), rnd(64, 64, 3, 3, seed=2, scale=0.06)
    g, be = 1 + 0.1 * rnd(64, seed=3), rnd(64, seed=4)
    up = rnd(B, (H // 2) * (W // 2), 64, seed=5) if H % 2 == 0 and W % 2 == 0 else None
    kw = dict(up=up.to(DEV), up_hw=(H // 2, W // 2)) if up is not None else {}
    xd = x.to(DEV)
    for big in (True, False):                                    # (a channel beyond the half range: the clamp, not infinities)
        gg = g.clone()
        if big:
            gg[0] = 3e5
        y32 = ops().groupnorm_tokens(xd, gg.to(DEV), be.to(DEV), H, W, **kw)
        y16 = ops().groupnorm_tokens(xd, gg.to(DEV), be.to(DEV), H, W, out_f16=True, **kw)
        assert y16.dtype == torch.float16 and y16.shape == y32.shape
        assert torch.equal(y16, y32.clamp(-65504.0, 65504.0).to(torch.float16)) and bool(torch.isfinite(y16).all())
        assert not big or H * W < 8 or float(y32.abs().max()) > 65504.0
    w3 = w.permute(0, 2, 3, 1).reshape(64, 576).contiguous().to(DEV)
    o_ref, st_ref = ops().conv3x3_c64(y32, w3, H, W, bf16="f16")
    o, st = ops().conv3x3_c64(y16, w3, H, W, bf16="f16")
    assert o.dtype == torch.float32 and torch.equal(o, o_ref)
    mom = torch.stack([o.double().sum(1), (o.double() ** 2).sum(1)], -1).cpu()
    torch.testing.assert_close(st.cpu(), mom, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(st.cpu(), st_ref.cpu(), rtol=1e-5, atol=1e-4)
    st0 = torch.ones(B, 64, 2, device=DEV, dtype=torch.float64)
    o2, st2 = ops().conv3x3_c64(y16, w3, H, W, bf16="f16", stats=st0, stats_cleared=True)
    assert torch.equal(o2, o) and st2.data_ptr() == st0.data_ptr()
    torch.testing.assert_close(st2.cpu(), mom + 1.0, rtol=1e-5, atol=1e-4)
    with pytest.raises(RuntimeError):
        ops().conv3x3_c64(y16, w3, H, W, bf16=True)


def test_kv_project_multi_equals_single_launches():
    """msm_kv_project_multi_f32: nine jobs (three levels x three layers, NCHW and token-major inputs) in one launch are
    bit-identical to nine msm_kv_project_f32 launches."""
    B, N = 8, 512
    buf = rnd(B, 30 * 40 + 60 * 80, 64, seed=50).to(DEV)                     # two levels as slices of one token buffer
    levels = [rnd(B, 64, 15, 20, seed=51).to(DEV),
              buf[:, :1200].view(B, 30, 40, 64).permute(0, 3, 1, 2), buf[:, 1200:].view(B, 60, 80, 64).permute(0, 3, 1, 2)]
    xs, ws, cs = [], [], []
    for i in range(9):
        x = levels[i % 3]
        xs.append(x)
        ws.append(rnd(N, 64, seed=60 + i, scale=0.1).to(DEV))
        cs.append(rnd(x.shape[2] * x.shape[3], N, seed=70 + i).to(DEV))
    outs = ops().kv_project_multi(xs, ws, cs)
    for x, w, c, o in zip(xs, ws, cs, outs):
        ref = torch.einsum("bchw,nc->bhwn", x.double(), w.double()).reshape(B, -1, N) + c.double()
        closed(o, ref.cpu(), rtol=2e-5, atol=2e-5)
        if x.shape[2] * x.shape[3] * B >= 8192 or ops().is_token_major(x):        # the single launch takes the same kernel there
            assert torch.equal(o, ops().kv_project(x, w, c))
    # low-precision mode: bf16 output.  KV_PIPE = 0: the same fp32 products, only the store rounded (round to nearest even);
    # default: bf16 MFMAs -- w rounded to one bf16, x as a hi + lo pair -- against float64 on those operands
    with option_ctx("KV_PIPE", 0):
        outs16 = ops().kv_project_multi(xs, ws, cs, out_dtype=torch.bfloat16)
    for o, o16 in zip(outs, outs16):
        assert o16.dtype == torch.bfloat16 and torch.equal(o16, o.to(torch.bfloat16))
    outs16 = ops().kv_project_multi(xs, ws, cs, out_dtype=torch.bfloat16)
    for x, w, c, o16 in zip(xs, ws, cs, outs16):
        ref = torch.einsum("bchw,nc->bhwn", x.double(), _bf16_round(w.cpu()).double().to(x.device)).reshape(B, -1, N) + c.double()
        err = (o16.double() - ref).abs().cpu()
        assert o16.dtype == torch.bfloat16 and float((err / (ref.abs().cpu() + 1.0)).max()) < 6e-3      # one bf16 rounding of the result (2^-8 relative)
    # precision "f16" (N = 512 = [K | V]): IEEE-half operands (one term each), the K columns stored as HALF bit patterns, V as bf16 --
    # against float64 on the fp16-rounded operands: K to one fp16 rounding of the result, V to one bf16 rounding
    if N == 512:
        outs_h = ops().kv_project_multi(xs, ws, cs, out_dtype=torch.bfloat16, keys_f16=True)
        h16 = lambda t: t.to(torch.float16).double()
        for x, w, c, oh, o in zip(xs, ws, cs, outs_h, outs):
            ref = torch.einsum("bchw,nc->bhwn", h16(x), h16(w)).reshape(B, -1, N) + c.double()
            kk = oh[..., :256].contiguous().view(torch.float16).double()
            vv = oh[..., 256:].double()
            assert float(((kk - ref[..., :256]).abs() / (ref[..., :256].abs() + 1.0)).max()) < 1.2e-3          # 2^-11 relative + the fp32 sum's rounding
            assert float(((vv - ref[..., 256:]).abs() / (ref[..., 256:].abs() + 1.0)).max()) < 6e-3
            # ... and against the EXACT fp32 result the K half is closer than a bf16 store could be
            assert float((kk - o[..., :256].double()).abs().mean()) < 0.25 * float((o[..., :256].to(torch.bfloat16).double() - o[..., :256].double()).abs().mean())
        with pytest.raises(RuntimeError, match="keys_f16"):
            ops().kv_project_multi(xs, ws, cs, keys_f16=True)
    # fp32 accuracy on the bf16 matrix pipe (exact three-term splits): the fp32 tolerances, and no further from float64 than the fp32 MFMAs
    outs_s = ops().kv_project_multi(xs, ws, cs, split=True)
    for x, w, c, o, os_ in zip(xs, ws, cs, outs, outs_s):
        ref = (torch.einsum("bchw,nc->bhwn", x.double(), w.double()).reshape(B, -1, N) + c.double()).cpu()
        closed(os_, ref, rtol=2e-5, atol=2e-5)
        e_s, e_m = (os_.double().cpu() - ref).abs().mean(), (o.double().cpu() - ref).abs().mean()
        assert float(e_s) <= 1.5 * float(e_m) + 1e-9


@pytest.mark.parametrize("B,H,W,Cout", [(2, 12, 16, 256), (1, 7, 36, 64), (1, 30, 40, 128)])
def test_conv3x3_c64_nchw_vs_fp64(B, H, W, Cout):
    """msm_conv3x3_c64_nchw_f32 (planar output, bias, several 64-channel slices) against an fp64 conv2d."""
    x, w, b = rnd(B, H * W, 64, seed=1), rnd(Cout, 64, 3, 3, seed=2, scale=0.06), rnd(Cout, seed=3)
    ref = F.conv2d(x.double().view(B, H, W, 64).permute(0, 3, 1, 2), w.double(), b.double(), padding=1).reshape(B, Cout, H * W)
    w3 = w.permute(0, 2, 3, 1).reshape(Cout, 576).contiguous().to(DEV)
    out = ops().conv3x3_tokens_to_nchw(x.to(DEV), w3, b.to(DEV), H, W)
    closed(out, ref, rtol=2e-5, atol=2e-5)
    closed(ops().conv3x3_tokens_to_nchw(x.to(DEV), w3, None, H, W), ref - b.double()[None, :, None], rtol=2e-5, atol=2e-5)
    # low-precision mode (msm_conv3x3_c64_nchw_bf16): against the fp64 convolution WITH THE ROUNDED WEIGHT to 1e-4 (the activations'
    # hi + lo pair carries 16 mantissa bits), against the unrounded one to bf16's tolerance
    ref_r = F.conv2d(x.double().view(B, H, W, 64).permute(0, 3, 1, 2), w.bfloat16().double(), b.double(), padding=1).reshape(B, Cout, H * W)
    lp = ops().conv3x3_tokens_to_nchw(x.to(DEV), w3, b.to(DEV), H, W, bf16=True)
    closed(lp, ref_r, rtol=1e-4, atol=1e-4)
    assert float((lp.cpu().double() - ref).abs().max()) < 2e-2 * float(ref.abs().max())
    # ... and with IEEE-half operands (msm_conv3x3_c64_nchw_f16): fp64 on the fp16-rounded operands
    h16 = lambda t: t.to(torch.float16).double()
    ref_h = F.conv2d(h16(x).view(B, H, W, 64).permute(0, 3, 1, 2), h16(w), b.double(), padding=1).reshape(B, Cout, H * W)
    lh = ops().conv3x3_tokens_to_nchw(x.to(DEV), w3, b.to(DEV), H, W, bf16="f16")
    closed(lh, ref_h, rtol=1e-4, atol=1e-4)
    assert 4 * float((lh.cpu().double() - ref).abs().mean()) <= float((lp.cpu().double() - ref).abs().mean())


@pytest.mark.parametrize("B,shapes", [(2, [(15, 20), (30, 40), (60, 80)]), (1, [(4, 6), (8, 12), (16, 24)]), (3, [(7, 7), (14, 14), (28, 28)])])
def test_msda_gather_with_fused_sampling_projection(B, shapes):
    """msm_msdeform_attn_enc_fused_fwd computes [sampling_offsets | attention_weights](src + pos) for its own queries on the
    matrix pipe instead of reading the tensor the token kernel wrote: same operands, same k order as msm_encoder_block_fwd, so
    the gather output is BITWISE the unfused path's (encoder block -> proj -> owner-record gather) -- and that path's gather
    kernels (owner records / round-2 / generic) agree with each other and with the oracle."""
    C, DF, H8 = 64, 1024, 8
    S = sum(h * w for h, w in shapes)
    ss = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    starts = _start(ss.cpu()).to(DEV)
    attn, src, pos = rnd(B, S, C, seed=1), rnd(B, S, C, seed=2), rnd(S, C, seed=3)
    wo, w1, w2 = rnd(C, C, seed=4, scale=C ** -0.5), rnd(DF, C, seed=6, scale=C ** -0.5), rnd(C, DF, seed=8, scale=DF ** -0.5)
    wv, wp, bp = rnd(C, C, seed=14, scale=C ** -0.5), rnd(288, C, seed=16, scale=0.5 * C ** -0.5), rnd(288, seed=17)
    d = lambda t: t.to(DEV).contiguous()
    stream = ops().pack_encoder_block(d(wo), d(w1), d(w2), d(wv), d(wp))
    small = torch.cat([rnd(C, seed=5, scale=0.1), 1 + 0.1 * rnd(C, seed=10), rnd(C, seed=11, scale=0.1), rnd(DF, seed=7, scale=0.1), rnd(C, seed=9, scale=0.1),
                       1 + 0.1 * rnd(C, seed=12), rnd(C, seed=13, scale=0.1), rnd(C, seed=15, scale=0.1), bp]).to(DEV)
    so, vh, po = ops().encoder_block(d(attn), d(src), stream, small, DF, 288, pos=d(pos), tokens_per_image=S, value_heads=H8)
    so2, vh2, po2 = ops().encoder_block(d(attn), d(src), stream, small, DF, 0, pos=d(pos), tokens_per_image=S, value_heads=H8)
    assert po2 is None and torch.equal(so, so2) and torch.equal(vh, vh2)          # proj_width = 0: the same block without the projection
    ref = ops().ms_deform_attn_encoder(vh, ss, starts, po, H8, 4)
    wpack, bpack = ops().pack_msda_proj(d(wp), d(bp), H8, 3, 4)
    got = ops().ms_deform_attn_encoder_fused(vh, ss, starts, so, d(pos), wpack, bpack, 4)
    assert torch.equal(got, ref)
    with option_ctx("MSDA_GENERIC", 2):        # round-2 gather kernel (IEEE divisions / expf in its prologue): same up to ~1e-6
        close(ops().ms_deform_attn_encoder(vh, ss, starts, po, H8, 4), ref.cpu(), rtol=1e-5, atol=2e-6)
    with option_ctx("MSDA_GENERIC", 1):
        close(ops().ms_deform_attn_encoder(vh, ss, starts, po, H8, 4), ref.cpu(), rtol=1e-5, atol=1e-6)
    # against the oracle's module arithmetic on the same projected values
    off = po.cpu()[..., :192].view(B, S, H8, 3, 4, 2)
    aw = torch.softmax(po.cpu()[..., 192:].view(B, S, H8, 12), -1).view(B, S, H8, 3, 4)
    refp = O.encoder_reference_points(shapes, B)                                  # (B,S,L,2)
    norm = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    loc = refp[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    value_tm = vh.cpu().permute(0, 2, 1, 3).contiguous()                          # (B,S,heads,8)
    want = O.ms_deform_attn_core(value_tm, shapes, loc.contiguous(), aw)
    close(got, want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("S,centres,noise,seed", [(100, 12, 0.02, 0), (100, 5, 0.25, 1), (300, 24, 0.3, 2), (304, 3, 0.35, 23), (37, 37, 0.0, 4),
                                                  (64, 1, 0.2, 5), (150, 8, 0.28, 6)])
def test_connected_components_on_device(S, centres, noise, seed):
    """msm_ms_connected_components (the order-dependent merge of mean_shift.py:41-76 on one wave) against the host loop and the
    oracle: identical labels, also where neighbourhoods overlap (noise levels that put many pairs around epsilon, so that the
    'take the mode of the labels already present' branch runs) and at the 304-seed limit."""
    from unseenobjectswithmeanshift_amd import mean_shift as ms
    g = torch.Generator().manual_seed(seed)
    c = F.normalize(torch.randn(centres, 64, generator=g), dim=1)
    Z = F.normalize(c[torch.randint(0, centres, (S,), generator=g)] + noise * torch.randn(S, 64, generator=g) / 8.0, dim=1)
    host = ms.connected_components_host(Z, 0.04)
    want = O.connected_components(Z, 0.04)
    assert torch.equal(host, want)
    # pairs within float rounding of the threshold could legitimately differ between summation orders: none here
    d = 0.5 * (1 - Z.double() @ Z.double().t())
    assert float((d - 0.04).abs().min()) > 1e-6
    got, num = ops().ms_connected_components(Z.to(DEV), 0.04)
    assert torch.equal(got.cpu(), want)
    assert int(num[1]) >= int(want.max()) + 1 and int(num[1]) <= S          # labels created
    assert int(num[0]) == int(torch.unique(want).numel()) == int(num[1])     # labels that survive (MS:211) = labels created
    assert torch.equal(ms.connected_components(Z.to(DEV), 0.04).cpu(), want)


def test_mean_shift_relabel_counts_only_the_first_num_labels():
    """mean_shift.py:211-222 takes the largest cluster among labels 0 .. len(unique(seed_labels)) - 1.  With label values that
    have a gap (here {0, 2}: num = 2, so label 2 is never counted) that differs from "the argmax over every label" -- the
    round-3 advisor finding.  (connected_components itself cannot leave a gap: the seed that opens a label is never inside a
    later seed's neighbourhood, the metric being symmetric -- asserted on the device pass in
    test_connected_components_on_device as num[0] == num[1] == len(unique).  The bound is honoured literally all the same.)"""
    e = torch.eye(64)
    Z = torch.stack([e[0], e[1], e[5]]).float()
    seed_labels = torch.tensor([0, 0, 2])                          # as if label 1 had vanished
    g = torch.Generator().manual_seed(0)
    X = F.normalize(torch.cat([e[0][None] + 0.01 * torch.randn(10, 64, generator=g), e[5][None] + 0.01 * torch.randn(50, 64, generator=g)]), dim=1)
    labels, counts = ops().ms_assign(X.to(DEV), Z.to(DEV), seed_labels.to(DEV), 4)
    ref = seed_labels[torch.argmin(0.5 * (1 - X @ Z.t()), dim=1)]
    assert torch.equal(labels.cpu(), ref) and counts.tolist() == [10, 0, 50, 0]
    num = torch.tensor([len(torch.unique(seed_labels))], dtype=torch.int32, device=DEV)
    out = ops().ms_relabel_largest_zero(labels.clone(), counts, num)
    assert torch.equal(out.cpu(), ref)                              # argmax over counts[:2] = label 0: nothing moves (the reference)
    out_all = ops().ms_relabel_largest_zero(labels.clone(), counts)
    assert out_all.cpu().tolist() == [2] * 10 + [0] * 50           # every label counted: the swap the reference does not make
    # the oracle's restatement of MS:206-229 on the same labels
    cnt = torch.tensor([(ref == i).sum() for i in range(int(num))])
    assert int(torch.argmax(cnt)) == 0


@pytest.mark.parametrize("B,Q,H,W,pool", [(2, 100, 16, 24, 2), (2, 100, 16, 24, 4), (8, 100, 120, 160, 8), (1, 100, 120, 160, 4),
                                          (2, 100, 120, 160, 2), (1, 300, 48, 64, 4), (2, 20, 8, 8, 2), (2, 100, 16, 24, 1), (1, 37, 18, 22, 2)])
def test_mask_logits_split_is_fp32_accurate(B, Q, H, W, pool):
    """msm_mask_logits_split_fwd (precision mode f32_split): the folded 64-channel mask step as six bf16 MFMAs per product on
    exact three-term splits of both operands.  An fp32 kernel in everything but the instruction it multiplies with: logits
    against the float64 einsum with an error no larger than 1.5x the fp32-MFMA kernel's (measured side by side, printed), the
    same attention-bit rules as test_mask_logits_folded_form."""
    C = 64
    wide = rnd(B, Q, 256, seed=1, scale=0.3)
    e, qb = wide[..., :C], wide[..., 64]
    f = rnd(B, C, H, W, seed=2)
    tgt = (H // pool, W // pool)
    full = torch.einsum("bqc,bchw->bqhw", e.double(), f.double()) + qb.double()[..., None, None]
    pooled = F.interpolate(full.float(), size=tgt, mode="bilinear", align_corners=False)
    attn_ref = pooled.sigmoid().flatten(2) < 0.5
    wd, fd = wide.to(DEV), f.to(DEV)
    packed = ops().pack_mask_features_split(fd)
    # the three terms reproduce the activation exactly: h + m + l == x in fp32
    terms = packed.view(torch.bfloat16).float()                                    # (B, 3, 8, HW, 8)
    back = (terms[:, 0] + terms[:, 1] + terms[:, 2]).permute(0, 1, 3, 2).reshape(B, C, H, W)
    assert torch.equal(back, fd)
    for want_mask, sparse in ((True, False), (False, False), (False, True)):
        mask, attn, row_any = ops().mask_logits(wd[..., :C], fd, want_mask=want_mask, target_size=tgt, sparse=sparse, qbias=wd[..., 64],
                                                packed_split=packed)
        if want_mask:
            m32, _, _ = ops().mask_logits(wd[..., :C], fd, want_mask=True, target_size=tgt, qbias=wd[..., 64])
            err_s = (mask.cpu().double() - full).abs()
            err_f = (m32.cpu().double() - full).abs()
            print(f"split mask step B={B} {H}x{W}: mean |err| {float(err_s.mean()):.3e} (fp32 MFMA {float(err_f.mean()):.3e}), "
                  f"max {float(err_s.max()):.3e} ({float(err_f.max()):.3e})")
            assert float(err_s.mean()) <= 1.5 * float(err_f.mean()) + 1e-9 and float(err_s.max()) <= 1.5 * float(err_f.max()) + 1e-7
            close(mask, full.float(), rtol=1e-4, atol=1e-4)
        got = attn.cpu().bool()
        diff = got != attn_ref
        if diff.any():
            assert pooled.flatten(2)[diff].abs().max() < 1e-4
        assert diff.float().mean() <= 1e-4
        assert torch.equal(row_any.cpu().bool(), ~attn.cpu().bool().all(-1))
    mask, attn, row_any = ops().mask_logits(wd[..., :C], fd, want_mask=True, target_size=None, qbias=wd[..., 64], packed_split=packed)
    close(mask, full.float(), rtol=1e-4, atol=1e-4)
    assert attn is None and row_any is None


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 48, 64), (1, 64, 7, 9), (2, 33, 10, 12), (1, 100, 8, 8)])
def test_l2_normalize_nchw(B, C, H, W):
    """msm_l2_normalize_nchw_f32 = F.normalize(x, p=2, dim=1) (pretrained_meanshiftformer_model.py:298-300), zero vectors included."""
    x = rnd(B, C, H, W, seed=1)
    x[0, :, 0, 0] = 0
    y = ops().l2_normalize_nchw(x.to(DEV))
    close(y, F.normalize(x, p=2, dim=1), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("N,Lq,M,L,P", [(2, 50, 8, 3, 4), (1, 7, 2, 2, 2), (1, 300, 8, 4, 4)])
def test_msda_locations_and_general_forward(N, Lq, M, L, P):
    """msm_msda_locations: softmax over L*P and loc = ref + off / (W_l, H_l) (ms_deform_attn.py:101-109) against torch."""
    shapes = torch.tensor([(6 + 3 * l, 4 + 5 * l) for l in range(L)], dtype=torch.int64)
    off, lg, ref = rnd(N, Lq, M, L, P, 2, seed=1), rnd(N, Lq, M, L * P, seed=2), torch.rand(N, Lq, L, 2, generator=torch.Generator().manual_seed(3))
    loc, aw = ops().msda_locations(off.to(DEV), lg.to(DEV), ref.to(DEV), shapes.to(DEV))
    norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    close(loc, ref[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :], rtol=1e-6, atol=1e-7)
    close(aw, torch.softmax(lg, -1).view(N, Lq, M, L, P), rtol=1e-5, atol=1e-7)
