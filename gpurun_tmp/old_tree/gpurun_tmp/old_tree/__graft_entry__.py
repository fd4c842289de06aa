"""Driver entry points: build() compiles every HIP source for gfx950 and imports the package;
smoke() runs one tiny invocation of the hot path on cuda:0 and checks it against the CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    import subprocess
    from unseenobjectswithmeanshift_amd import build as b
    out = b.build()
    # the oracle's C restatement (checker only; building it is not using it)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    from unseenobjectswithmeanshift_amd import _lib
    L = _lib.lib()
    missing = [s for s in _lib.declared_symbols() if not hasattr(L, s)]
    if missing:
        raise RuntimeError(f"{out} does not export {missing}")
    import unseenobjectswithmeanshift_amd.modeling  # noqa: F401
    import unseenobjectswithmeanshift_amd.meta_arch  # noqa: F401
    import unseenobjectswithmeanshift_amd.mean_shift  # noqa: F401
    print(f"built {out}: abi {L.msm_abi_version()}, {len(_lib.declared_symbols())} symbols")


def smoke() -> None:
    """Tiny end-to-end pass on cuda:0: pixel decoder -> 9-layer hypersphere decoder -> instance
    post-processing on a 64x96 frame, plus one mean-shift clustering, each checked against the
    oracle (oracle/ is the checker here, never the thing that runs the product path)."""
    import torch
    from oracle import msm_oracle as O
    from unseenobjectswithmeanshift_amd import synthetic as syn
    from unseenobjectswithmeanshift_amd import mean_shift as ms
    from unseenobjectswithmeanshift_amd.meta_arch import MeanShiftMaskFormer, build_resnet50_head

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = "cuda:0"
    head = build_resnet50_head()
    pd_sd = syn.synth_state_dict(syn.pixel_decoder_param_shapes())
    dec_sd = syn.synth_state_dict(syn.decoder_param_shapes())
    head.pixel_decoder.load_state_dict(pd_sd, strict=True)
    head.predictor.load_state_dict(dec_sd, strict=True)
    head = head.to(dev).eval()
    model = MeanShiftMaskFormer(backbone=None, sem_seg_head=head, num_queries=100)
    feats = syn.synth_backbone_features(2, 64, 96, seed=3)
    scores, classes, masks, boxes, qidx = model.inference({k: v.to(dev) for k, v in feats.items()}, (64, 96))
    torch.cuda.synchronize()
    mf, _, ms_feats = O.pixel_decoder_forward(pd_sd, feats)
    ref = O.decoder_forward(dec_sd, ms_feats, mf)
    out, _ = head({k: v.to(dev) for k, v in feats.items()})
    err_logits = (out["pred_logits"].cpu() - ref["pred_logits"]).abs().max().item()
    err_masks = (out["pred_masks"].cpu() - ref["pred_masks"]).abs().max().item()
    flips = ((out["pred_masks"].cpu() > 0) != (ref["pred_masks"] > 0)).float().mean().item()
    print(f"smoke: decoder max|dlogits|={err_logits:.2e} max|dmask|={err_masks:.2e} sign flips={flips:.2e}")
    assert err_logits < 1e-3 and err_masks < 5e-3 and flips < 1e-3
    assert masks.shape == (2, 20, 64, 96) and torch.isfinite(scores).all()
    X, ids = syn.synth_unit_embeddings(4800, 64, clusters=8, sigma=0.15, seed=18)
    labels, sel = ms.mean_shift_smart_init(X.to(dev), kappa=20, num_seeds=50, max_iters=10, first_index=7)
    ref_labels, ref_sel, _, _ = O.mean_shift_smart_init(X, 20.0, 50, 10, 7)
    agree = (labels.cpu() == ref_labels).float().mean().item()
    print(f"smoke: mean-shift label agreement={agree:.4f}, seeds equal={bool(torch.equal(sel.cpu(), ref_sel))}")
    assert agree > 0.999
    print("smoke OK")


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
