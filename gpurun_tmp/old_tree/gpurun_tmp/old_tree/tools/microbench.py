#!/usr/bin/env python
"""Kernel micro-benchmarks on one GPU (tuning aid, not part of the product or the tests).
   python tools/microbench.py gemm | enc | attn | mask"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unseenobjectswithmeanshift_amd import _lib, ops  # noqa: E402

if os.environ.get("MSM_LIB"):        # experiment builds of the library (tuning only)
    _lib.LIB_PATH = os.environ["MSM_LIB"]

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3   # us


def timeit_graph(fn, reps=20, iters=10):
    """GPU time of one call with the host out of the way: `reps` calls captured in a HIP graph, replayed."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        g.replay()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / (iters * reps) * 1e3   # us


def gemm():
    shapes = [("ffn1", 50400, 1024, 64), ("ffn2", 50400, 64, 1024), ("val", 50400, 64, 64), ("proj", 50400, 288, 64),
              ("kv2", 38400, 256, 256), ("kv1", 9600, 256, 256), ("q", 800, 256, 256), ("dffn1", 800, 2048, 256)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(N, K, device=DEV) * K ** -0.5
        b = torch.randn(N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        t = timeit(lambda: ops.gemm(a, w, b, out=out))
        print(f"{name:6s} M={M:6d} N={N:5d} K={K:5d}  {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TFLOP/s  "
              f"tile={_lib.lib().msm_get_option(_lib.OPTIONS.index('GEMM_TILE'))}", flush=True)


def mask():
    """Mask step at B=8, 120x160 features: the literal contraction (C=256) and the folded one (C=64 activation, embedding =
    leading 64 columns of a 256-wide buffer, per-query bias) -- HIP-graph timed."""
    for C in (256, 64):
        wide = torch.randn(8, 100, 256, device=DEV) * 0.3
        e = wide[..., :C]
        qb = wide[..., 64] if C == 64 else None
        f = torch.randn(8, C, 120, 160, device=DEV)
        flops = 2.0 * 100 * C * 19200 * 8
        for nc, occ in (("", 5), ("", -1)):       # kernel 5: without the 4-query block on the 4x4x1 MFMA
            _lib.set_option("MASK_NC", int(nc) if nc else _lib.OPT_AUTO)
            _lib.set_option("MASK_KERNEL", occ)
            for tgt, wm in (((15, 20), False), ((30, 40), False), ((60, 80), False), (None, True)):
                t = timeit_graph(lambda: ops.mask_logits(e, f, want_mask=wm, target_size=tgt, qbias=qb))
                print(f"mask C={C} nc={nc or 'auto'} kernel={occ} target={tgt} write={wm}: {t:7.1f} us  {flops / t / 1e6:6.1f} TFLOP/s executed", flush=True)


def maskbf16():
    """bf16 mask step (configs 3/5): HBM-bound stream over the packed feature map."""
    e = torch.randn(8, 100, 256, device=DEV) * 0.3
    f = torch.randn(8, 256, 120, 160, device=DEV)
    t = timeit_graph(lambda: ops.pack_mask_features_bf16(f))
    print(f"pack_mask_features_bf16: {t:7.1f} us  {(f.numel() * 6) / t / 1e6:5.2f} TB/s", flush=True)
    fp = ops.pack_mask_features_bf16(f)
    flops = 2.0 * 100 * 256 * 19200 * 8
    for tgt, wm in (((15, 20), False), ((30, 40), False), ((60, 80), False), (None, True)):
        t = timeit_graph(lambda: ops.mask_logits(e, f, want_mask=wm, target_size=tgt, packed_bf16=fp))
        by = fp.numel() * 2 + e.numel() * 4 + (8 * 100 * 19200 * 4 if wm else 8 * 100 * tgt[0] * tgt[1])
        t32 = timeit_graph(lambda: ops.mask_logits(e, f, want_mask=wm, target_size=tgt))
        print(f"mask bf16 target={tgt} write={wm}: {t:7.1f} us  {by / t / 1e6:5.2f} TB/s  {flops / t / 1e6:6.1f} TFLOP/s   (fp32: {t32:6.1f} us)",
              flush=True)


def enc():
    B, S = 8, 6300
    attn, src, pos = torch.randn(B, S, 64, device=DEV), torch.randn(B, S, 64, device=DEV), torch.randn(S, 64, device=DEV)
    wo, w1, w2 = torch.randn(64, 64, device=DEV) * .1, torch.randn(1024, 64, device=DEV) * .1, torch.randn(64, 1024, device=DEV) * .03
    wv, wp = torch.randn(64, 64, device=DEV) * .1, torch.randn(288, 64, device=DEV) * .1
    stream = ops.pack_encoder_block(wo, w1, w2, wv, wp)
    small = torch.randn(64 * 7 + 1024 + 288, device=DEV) * .1
    t = timeit(lambda: ops.encoder_block(attn, src, stream, small, 1024, 288, pos=pos, tokens_per_image=S), iters=30)
    fl = B * S * 2.0 * (64 * 64 * 2 + 64 * 1024 * 2 + 64 * 288)
    print(f"enc_block: {t:7.1f} us  {fl / t / 1e6:6.1f} TFLOP/s", flush=True)
    value, proj = torch.randn(B, S, 64, device=DEV), torch.randn(B, S, 288, device=DEV)
    ss = torch.tensor([(15, 20), (30, 40), (60, 80)], dtype=torch.int64, device=DEV)
    st = torch.tensor([0, 300, 1500], dtype=torch.int64, device=DEV)
    t = timeit(lambda: ops.ms_deform_attn_encoder(value, ss, st, proj, 8, 4), iters=30)
    print(f"msda enc: {t:7.1f} us", flush=True)
    vhm = ops.value_to_head_major(value, 8)
    proj.mul_(0.3)                                   # sampling offsets of a few pixels, like a trained model's
    t = timeit_graph(lambda: ops.ms_deform_attn_encoder(vhm, ss, st, proj, 8, 4))
    print(f"msda enc, head-major value (quad-cooperative D=8 kernel): {t:7.1f} us", flush=True)
    with _lib.option("MSDA_GENERIC", 1):
        t = timeit_graph(lambda: ops.ms_deform_attn_encoder(vhm, ss, st, proj, 8, 4))
    print(f"msda enc, head-major value (generic kernel): {t:7.1f} us", flush=True)
    # same taps per lane and bytes, but 64-byte instead of 32-byte contiguous segments (4 heads x 16 dims)
    proj4 = torch.randn(B, S, 144, device=DEV)
    t = timeit(lambda: ops.ms_deform_attn_encoder(value, ss, st, proj4, 4, 4), iters=30)
    print(f"msda enc, 4 heads x 16 dims (segment-size experiment): {t:7.1f} us", flush=True)


def attn():
    B, E = 8, 256
    for S in (300, 1200, 4800, 100):
        q, k, v = torch.randn(B, 100, E, device=DEV), torch.randn(B, S, 2 * E, device=DEV), None
        m = (torch.rand(B, 100, S, device=DEV) < 0.5).to(torch.uint8)
        ra = torch.ones(B, 100, device=DEV, dtype=torch.int32)
        t = timeit_graph(lambda: ops.hypersphere_attention(q, k[..., :E], k[..., E:], 8, masked=m, row_any=ra))
        fl = 2.0 * 2 * B * 100 * S * E
        t0 = timeit_graph(lambda: ops.hypersphere_attention(q, k[..., :E], k[..., E:], 8))
        print(f"hs_attn S={S}: {t:7.1f} us  {fl / t / 1e6:6.1f} TFLOP/s   (without a mask: {t0:6.1f} us)", flush=True)


def kv():
    """Folded K/V projection of the three feature levels (conv1x1_nchw_to_tokens with a matrix bias), B=8."""
    B, Cin, E = 8, 64, 256
    for (h, w) in ((15, 20), (30, 40), (60, 80)):
        x = torch.randn(B, Cin, h, w, device=DEV)
        wt = torch.randn(2 * E, Cin, device=DEV) * 0.1
        c = torch.randn(h * w, 2 * E, device=DEV)
        t = timeit_graph(lambda: ops.conv1x1_nchw_to_tokens(x, wt, c))
        by = (x.numel() + c.numel() + B * h * w * 2 * E) * 4
        print(f"kv proj {h}x{w}: {t:6.1f} us  {2.0 * B * h * w * Cin * 2 * E / t / 1e6:6.1f} TFLOP/s  {by / t / 1e6:6.2f} TB/s",
              flush=True)
        t2 = timeit_graph(lambda: ops.kv_project(x, wt, c))
        err = (ops.kv_project(x, wt, c) - ops.conv1x1_nchw_to_tokens(x, wt, c)).abs().max().item()
        print(f"   kv_project kernel: {t2:6.1f} us  {by / t2 / 1e6:6.2f} TB/s  max|diff| {err:.2e}", flush=True)


def convs():
    """Pixel-decoder front end at B=8, 640x480: the 1x1 input projections (NCHW -> tokens) + GroupNorm passes, the
    layer-0 value / sampling projections and the 3x3 convolution."""
    B = 8
    for name, cin, h, w in (("res5", 2048, 15, 20), ("res4", 1024, 30, 40), ("res3", 512, 60, 80), ("res2", 256, 120, 160)):
        x = torch.randn(B, cin, h, w, device=DEV)
        wt = torch.randn(64, cin, device=DEV) * 0.05
        b = torch.randn(64, device=DEV)
        g, be = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
        t = timeit_graph(lambda: ops.conv1x1_nchw_to_tokens(x, wt, b))
        tok = ops.conv1x1_nchw_to_tokens(x, wt, b)
        t2 = timeit_graph(lambda: ops.groupnorm_tokens(tok, g, be, h, w, groups=32))
        st0 = torch.zeros(B, 64, 2, device=DEV, dtype=torch.float64)
        wpk = ops.pack_conv_in_weight(wt)
        for ntv in ("1", "2", "4", ""):
            _lib.set_option("CONVIN_NT", int(ntv) if ntv else _lib.OPT_AUTO)
            t3 = timeit_graph(lambda: ops.conv1x1_in(x, wpk, b, stats=st0, stats_cleared=True))
            o3, s3 = ops.conv1x1_in(x, wpk, b)
            err = (o3 - tok).abs().max().item()
            serr = ((s3 - ops.groupnorm_stats(tok)).abs() / (ops.groupnorm_stats(tok).abs() + 1)).max().item()
            print(f"   conv1x1_in NT={ntv or 'auto'} (+GroupNorm moments, no fill): {t3:6.1f} us   max|diff| {err:.2e}  moments rel {serr:.2e}",
                  flush=True)
        by = (x.numel() + tok.numel()) * 4
        print(f"{name} {cin}->64 @{h}x{w}: conv {t:6.1f} us ({by / t / 1e6:5.2f} TB/s, {2.0 * B * h * w * cin * 64 / t / 1e6:5.1f} TFLOP/s)"
              f"   groupnorm {t2:6.1f} us", flush=True)
    _lib.set_option("CONVIN_NT", _lib.OPT_AUTO)
    xs = [torch.randn(B, c, h, w, device=DEV) for c, h, w in ((2048, 15, 20), (1024, 30, 40), (512, 60, 80))]
    wps = [ops.pack_conv_in_weight(torch.randn(64, x.shape[1], device=DEV) * 0.05) for x in xs]
    bs = [torch.randn(64, device=DEV) for _ in xs]
    buf = torch.empty(B, 6300, 64, device=DEV)
    st3 = torch.zeros(3, B, 64, 2, device=DEV, dtype=torch.float64)
    print(f"conv1x1_in_multi res5+res4+res3 (137 MB): {timeit_graph(lambda: ops.conv1x1_in_multi(xs, wps, bs, buf, st3, stats_cleared=True)):6.1f} us",
          flush=True)
    gnp = torch.stack([torch.stack([torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)]) for _ in xs]).contiguous()
    pstream = ops.pack_encoder_prologue(torch.randn(64, 64, device=DEV) * 0.1, torch.randn(288, 64, device=DEV) * 0.1)
    psmall = torch.randn(64 + 288, device=DEV)
    ppos = torch.randn(6300, 64, device=DEV)
    ops.conv1x1_in_multi(xs, wps, bs, buf, st3, stats_cleared=True)
    print(f"encoder_prologue (GroupNorm + value / sampling projections of layer 0, 97 MB): "
          f"{timeit_graph(lambda: ops.encoder_prologue(buf, st3, gnp, [0, 300, 1500, 6300], pstream, psmall, ppos, 288, value_heads=8)):6.1f} us", flush=True)
    src = torch.randn(B, 6300, 64, device=DEV)
    wv, bv = torch.randn(64, 64, device=DEV) * 0.1, torch.randn(64, device=DEV)
    wp, bp = torch.randn(288, 64, device=DEV) * 0.1, torch.randn(288, device=DEV)
    pos = torch.randn(6300, 64, device=DEV)
    print(f"value_proj: {timeit_graph(lambda: ops.gemm(src, wv, bv)):6.1f} us   value_to_head_major: "
          f"{timeit_graph(lambda: ops.value_to_head_major(src, 8)):6.1f} us   offsets/weights proj: "
          f"{timeit_graph(lambda: ops.gemm(src, wp, bp, a2=pos)):6.1f} us", flush=True)
    y = torch.randn(B, 19200, 64, device=DEV)
    w3 = torch.randn(64, 576, device=DEV) * 0.05
    t = timeit_graph(lambda: ops.conv3x3_tokens(y, w3, 120, 160))
    t3 = timeit_graph(lambda: ops.conv3x3_c64(y, w3, 120, 160))
    t3b = timeit_graph(lambda: ops.conv3x3_c64(y, w3, 120, 160, bf16=True))
    print(f"conv3x3_c64 bf16 mode: {t3b:6.1f} us", flush=True)
    o3, s3 = ops.conv3x3_c64(y, w3, 120, 160)
    ref3 = ops.conv3x3_tokens(y, w3, 120, 160)
    print(f"conv3x3_c64 (weight in LDS, + GroupNorm moments incl. fill): {t3:6.1f} us ({2.0 * B * 19200 * 576 * 64 / t3 / 1e6:5.1f} TFLOP/s)  "
          f"max|diff| {(o3 - ref3).abs().max().item():.2e}  moments rel {((s3 - ops.groupnorm_stats(ref3)).abs() / (ops.groupnorm_stats(ref3).abs() + 1)).max().item():.2e}", flush=True)
    print(f"conv3x3 64->64 @120x160: {t:6.1f} us ({2.0 * B * 19200 * 576 * 64 / t / 1e6:5.1f} TFLOP/s)   groupnorm_stats "
          f"{timeit_graph(lambda: ops.groupnorm_stats(y)):6.1f} us", flush=True)


def twostage():
    """BASELINE configs[3]: two-stage RGB + depth-crop refinement at 640x480 over 16 frames (first stage on the frame,
    depth filter, ROI crops resized to 224x224, one BATCHED second stage over all crops, paste-back).  The backbone is out of
    scope: a cheap stand-in (tests/test_gpu_modules._TinyBackbone) produces res2..res5."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import test_gpu_modules as tg
    from unseenobjectswithmeanshift_amd import two_stage as ts
    from unseenobjectswithmeanshift_amd.meta_arch import Instances, MeanShiftMaskFormer, Network_RGBD
    head = tg.make_pixel_decoder()
    bb = tg._TinyBackbone().to(DEV).eval()

    class RGBD(MeanShiftMaskFormer):
        def forward(self, batched_inputs):
            imgs = torch.stack([x["image"] for x in batched_inputs])
            deps = torch.stack([x["depth"] for x in batched_inputs])
            H, W = imgs.shape[-2:]
            feats = self.backbone(imgs, deps)
            if os.environ.get("MSM_TS_GRAPH"):          # HIP-graph replay per geometry (frame / number of crops)
                if getattr(self, "_g", None) is None:
                    self._g = self.graphed()
                scores, classes, masks, boxes, _ = self._g(feats, (int(H), int(W)))
            else:
                scores, classes, masks, boxes, _ = self.inference(feats, (int(H), int(W)))
            return [{"instances": Instances((int(H), int(W)), pred_masks=masks[b], pred_boxes=boxes[b], scores=scores[b],
                                            pred_classes=classes[b])} for b in range(len(batched_inputs))]

    model = RGBD(backbone=bb, sem_seg_head=head, num_queries=100)
    crops = []

    class Pred(Network_RGBD):
        def batch_call(self, samples):
            crops.append(len(samples))
            with torch.no_grad():
                return self.model(samples)

    first, second = Network_RGBD(model), Pred(model)
    g = torch.Generator().manual_seed(3)
    frames = [(torch.rand(3, 480, 640, generator=g).to(DEV), torch.rand(3, 480, 640, generator=g).to(DEV)) for _ in range(16)]

    def run():
        for im, dp in frames:
            ts.test_sample_crop_nolabel({"image_color": im, "depth": dp}, first, second, confident_score=0.0, topk=False)

    run()
    torch.cuda.synchronize()
    crops.clear()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if os.environ.get("MSM_CPROFILE"):                     # host-side breakdown of one more pass
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        run()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats(os.environ["MSM_CPROFILE"] if os.environ["MSM_CPROFILE"] in ("tottime", "cumulative") else "cumulative").print_stats(45)
    print(f"two-stage 640x480, 16 frames: {dt * 1e3 / 16:7.2f} ms per frame = {16 / dt:6.1f} frames/s, "
          f"{sum(crops) / max(1, len(crops)):.1f} crops per frame in one batched second-stage call", flush=True)


def latency():
    """Single-frame latency of the hot path (B=1, 640x480): eager Python launches against HIP-graph replay."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import test_gpu_modules as tg
    from unseenobjectswithmeanshift_amd import synthetic as syn
    from unseenobjectswithmeanshift_amd.meta_arch import MeanShiftMaskFormer
    model = MeanShiftMaskFormer(backbone=None, sem_seg_head=tg.make_pixel_decoder(), num_queries=100)
    g = model.graphed()
    for B in (1, 2, 4, 8):
        feats = {k: v.to(DEV) for k, v in syn.synth_backbone_features(B, 480, 640, seed=3).items()}
        te = timeit(lambda: model.inference(feats, (480, 640)), iters=20)
        tg_ = timeit(lambda: g(feats, (480, 640)), iters=20)
        print(f"B={B}: eager {te / 1e3:6.2f} ms ({B / te * 1e6:7.1f} images/s)   graph replay incl. input copy {tg_ / 1e3:6.2f} ms "
              f"({B / tg_ * 1e6:7.1f} images/s)", flush=True)


def tails():
    """Fused decoder-layer tails (csrc/dec_chain.hip) at B=8, Q=100."""
    B, Q, E, Fh = 8, 100, 256, 2048
    r = lambda *s: torch.randn(*s, device=DEV)
    o, res, qpos = r(B, Q, E), r(B, Q, E), r(Q, E)
    pk = lambda n, k: ops.dec_pack_weight(r(n, k) * k ** -0.5)
    wo, w_in, w1, w2, wq = pk(E, E), pk(3 * E, E), pk(Fh, E), pk(E, Fh), pk(E, E)
    mlp = [(pk(E, E), r(E)) for _ in range(3)]
    v = lambda: r(E)
    bo, g, b, b_in, b1, b2 = v(), v(), v(), r(3 * E), r(Fh), v()
    t = timeit_graph(lambda: ops.dec_post_cross(o, res, qpos, wo, bo, g, b, w_in, b_in))
    print(f"dec_post_cross: {t:6.1f} us  (2 stages/block)", flush=True)
    for n_parts in (8, 4, 2):
        t = timeit_graph(lambda: ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2, n_parts=n_parts))
        print(f"dec_post_self n_parts={n_parts}: {t:6.1f} us  ({1 + 2 * 8 // n_parts} stages/block)", flush=True)
    x, parts = ops.dec_post_self(o, res, wo, bo, g, b, w1, b1, w2)
    t = timeit_graph(lambda: ops.dec_heads(x, g, b, mlp, parts=parts, bias=b2, ln_g=g, ln_b=b, l2norm=True, wq=wq, bq=bo,
                                     query_pos=qpos))
    print(f"dec_heads: {t:6.1f} us  (3 stages/block)", flush=True)


def ucn():
    """RGB-D / UCN configuration at full size: 307 200 keys per image, 6 decoder layers."""
    from unseenobjectswithmeanshift_amd import synthetic as syn
    from unseenobjectswithmeanshift_amd.meta_arch import MeanShiftMaskFormer, build_ucn_head
    B = int(os.environ.get("UCN_B", "2"))
    head = build_ucn_head()
    head.pixel_decoder.load_state_dict(syn.synth_state_dict({"mask_features.weight": (256, 64, 3, 3), "mask_features.bias": (256,)}, salt=3))
    head.predictor.load_state_dict(syn.synth_state_dict(syn.decoder_param_shapes(dec_layers=6, num_feature_levels=1), salt=4))
    head = head.to(DEV).eval()
    model = MeanShiftMaskFormer(backbone=None, sem_seg_head=head, num_queries=100)
    X, _ = syn.synth_unit_embeddings(B * 480 * 640, 64, clusters=12, sigma=0.3, seed=5)
    feat = {"res5": X.view(B, 480 * 640, 64).transpose(1, 2).reshape(B, 64, 480, 640).contiguous().to(DEV)}
    t = timeit(lambda: model.inference(feat, (480, 640)), iters=5, warm=2)
    print(f"ucn B={B} 480x640: {t / 1e3:8.2f} ms per batch, {B / (t * 1e-6):7.1f} images/s, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    # with the RGB-D backbone in front (two ResNet34-8s towers through MIOpen, BatchNorm folded)
    from unseenobjectswithmeanshift_amd.meta_arch import PretrainedMeanShiftMaskFormer
    from unseenobjectswithmeanshift_amd.ucn_backbone import UCNBackbone
    bb = UCNBackbone()
    bb.load_state_dict(syn.ucn_backbone_state_dict(salt=6))
    full = PretrainedMeanShiftMaskFormer(backbone=bb.to(DEV).eval(), sem_seg_head=head, num_queries=100)
    img, dep = torch.randn(B, 3, 480, 640, device=DEV), torch.randn(B, 3, 480, 640, device=DEV)
    tb = timeit(lambda: bb(img, None, dep), iters=5, warm=3)
    t = timeit(lambda: full([{"image": img, "depth": dep}]), iters=5, warm=2)
    print(f"ucn end to end (backbone {tb / 1e3:.2f} ms + head): {t / 1e3:8.2f} ms per batch, {B / (t * 1e-6):7.1f} images/s", flush=True)


def cfg5():
    """BASELINE configs[4]: 1280x960, 300 queries, 20 decoder layers (19 + heads), B=1."""
    from unseenobjectswithmeanshift_amd import synthetic as syn
    from unseenobjectswithmeanshift_amd.meta_arch import MeanShiftMaskFormer, build_resnet50_head
    head = build_resnet50_head(num_queries=300, dec_layers=19)
    head.pixel_decoder.load_state_dict(syn.synth_state_dict(syn.pixel_decoder_param_shapes()))
    head.predictor.load_state_dict(syn.synth_state_dict(syn.decoder_param_shapes(num_queries=300, dec_layers=19)))
    model = MeanShiftMaskFormer(backbone=None, sem_seg_head=head.to(DEV).eval(), num_queries=300)
    for B in (1, 4):
        feats = {k: v.to(DEV) for k, v in syn.synth_backbone_features(B, 960, 1280, seed=9).items()}
        for mode in ("f32", "bf16"):
            head.predictor.mask_step_dtype = mode
            t = timeit(lambda: model.inference(feats, (960, 1280)), iters=5, warm=2)
            print(f"cfg5 1280x960 Q=300 L=19 B={B} mask step {mode}: {t / 1e3:8.2f} ms per batch, {B / (t * 1e-6):7.1f} images/s, "
                  f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


def meanshift():
    """Classic UCN clustering at 640x480 (n = 307200, S = 100, 10 iterations) and the cfg-5 stress size."""
    from unseenobjectswithmeanshift_amd import mean_shift as ms
    from unseenobjectswithmeanshift_amd import synthetic as syn
    for n, S, iters, k in ((307200, 100, 10, 12), (1228800, 300, 20, 24)):
        X, _ = syn.synth_unit_embeddings(n, 64, clusters=k, sigma=0.15, seed=3)
        Xd = X.to(DEV)
        t_seed = timeit(lambda: ops.ms_select_seeds(Xd, S, 7), iters=3, warm=1)
        seeds, _ = ops.ms_select_seeds(Xd, S, 7)
        t_hill = timeit(lambda: ops.ms_hill_climb(Xd, seeds, 20.0, iters), iters=3, warm=1)
        Z = ops.ms_hill_climb(Xd, seeds, 20.0, iters)
        lab = torch.zeros(S, dtype=torch.int64, device=DEV)
        t_asg = timeit(lambda: ops.ms_assign(Xd, Z, lab, 1), iters=3, warm=1)
        for _ in range(2):                      # warm-up: host BLAS threads, allocator
            ms.mean_shift_smart_init(Xd, 20.0, S, iters, first_index=7)
        torch.cuda.synchronize()
        reps = 10 if n < 500000 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            labels, sel = ms.mean_shift_smart_init(Xd, 20.0, S, iters, first_index=7)
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / reps
        print(f"mean-shift n={n} S={S} it={iters}: seeding {t_seed / 1e3:7.2f} ms ({S * n * 256 / t_seed / 1e6:6.2f} TB/s), "
              f"hill-climb {t_hill / 1e3:7.2f} ms ({4.0 * S * n * 64 * iters / t_hill / 1e6:6.1f} TFLOP/s), assign {t_asg / 1e3:6.2f} ms, "
              f"end-to-end {t_all * 1e3:7.2f} ms = {1 / t_all:6.1f} images/s, clusters={int(labels.max()) + 1}", flush=True)


if __name__ == "__main__":
    {"gemm": gemm, "mask": mask, "enc": enc, "attn": attn, "ucn": ucn, "meanshift": meanshift, "cfg5": cfg5,
     "tails": tails, "kv": kv, "maskbf16": maskbf16, "twostage": twostage, "convs": convs, "latency": latency}[sys.argv[1]]()
