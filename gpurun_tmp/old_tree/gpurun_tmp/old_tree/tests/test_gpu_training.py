"""f3: decoder backward on the HIP path (pytest -m gpu): the hypersphere-attention gradient kernel against float64 autograd
of the oracle, and the differentiable decoder (training.decoder_forward_train) against the gradients of the imported reference
decoder (tests/golden/decoder_backward.npz, float64 autograd)."""
import numpy as np
import pytest
import torch

from oracle import msm_oracle as O
from unseenobjectswithmeanshift_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("B,Lq,S,masked", [(2, 100, 300, True), (1, 100, 100, False), (2, 37, 1200, True), (1, 5, 7, True)])
def test_hypersphere_attention_backward_vs_fp64_autograd(B, Lq, S, masked):
    from unseenobjectswithmeanshift_amd import ops
    H, E = 8, 256
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B, n, E, generator=g) for n in (Lq, S, S))
    gout = torch.randn(B, Lq, E, generator=g)
    m = row_any = None
    if masked:
        m = torch.rand(B, Lq, S, generator=g) < 0.5
        m[:, 0] = True                                   # an all-masked row: attends everywhere (DEC:618)
        m[:, 1:, 0] = False
        row_any = (~m).any(-1).to(torch.int32)
    # float64 reference: the oracle's attention per head under autograd
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    add = None
    if masked:
        eff = m & (row_any[..., None] != 0)
        add = torch.zeros(B, 1, Lq, S, dtype=torch.float64).masked_fill(eff[:, None], float("-inf"))

    def heads(t, n):
        return t.view(B, n, H, 32).transpose(1, 2)
    logits = 30.0 * torch.matmul(torch.nn.functional.normalize(heads(qd, Lq), dim=-1, eps=1e-12),
                                 torch.nn.functional.normalize(heads(kd, S), dim=-1, eps=1e-12).transpose(-2, -1))
    if add is not None:
        logits = logits + add
    o = torch.matmul(torch.softmax(logits, -1), heads(vd, S))
    out = torch.nn.functional.normalize(o, dim=-1, eps=1e-12).transpose(1, 2).reshape(B, Lq, E)
    out.backward(gout.double())
    args = dict(masked=None if m is None else m.to(torch.uint8).to(DEV), row_any=None if row_any is None else row_any.to(DEV))
    fwd = ops.hypersphere_attention(q.to(DEV), k.to(DEV), v.to(DEV), H, **args)
    torch.testing.assert_close(fwd.cpu().double(), out.detach(), rtol=1e-4, atol=2e-5)
    gq, gk, gv = ops.hypersphere_attention_backward(q.to(DEV), k.to(DEV), v.to(DEV), H, gout.to(DEV), **args)
    for got, ref, name in ((gq, qd.grad, "q"), (gk, kd.grad, "k"), (gv, vd.grad, "v")):
        scale = float(ref.abs().max())
        err = float((got.cpu().double() - ref).abs().max())
        assert err < 2e-4 * scale + 1e-7, (name, err, scale)


def test_linear_and_mask_step_functions_vs_autograd():
    from unseenobjectswithmeanshift_amd import training as tr
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 100, 256, generator=g)
    w, b = torch.randn(768, 256, generator=g) * 0.05, torch.randn(768, generator=g)
    gy = torch.randn(2, 100, 512, generator=g)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = tr.linear(xd, wd[256:], bd[256:])                                      # a row slice of a packed in-projection weight
    y.backward(gy.to(DEV))
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    y64 = torch.nn.functional.linear(x64, w64[256:], b64[256:])
    y64.backward(gy.double())
    for got, ref in ((y, y64), (xd.grad, x64.grad), (wd.grad, w64.grad), (bd.grad, b64.grad)):
        torch.testing.assert_close(got.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-4)
    e, f = torch.randn(2, 100, 256, generator=g) * 0.3, torch.randn(2, 256, 16, 24, generator=g)
    gm = torch.randn(2, 100, 16, 24, generator=g)
    ed, fd = e.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
    mask, attn, row_any = tr._MaskStep.apply(ed, fd, (8, 12))
    assert attn.shape == (2, 100, 96) and not attn.requires_grad
    mask.backward(gm.to(DEV))
    e64, f64 = e.double().requires_grad_(True), f.double().requires_grad_(True)
    m64 = torch.einsum("bqc,bchw->bqhw", e64, f64)
    m64.backward(gm.double())
    for got, ref in ((mask, m64), (ed.grad, e64.grad), (fd.grad, f64.grad)):
        torch.testing.assert_close(got.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=2e-4)


def test_decoder_backward_vs_reference(golden):
    """Gradients of the fixed random functional of all ten predictions (make_golden.decoder_backward_loss) through the HIP
    decoder against float64 autograd through the imported reference decoder: the loss, the input gradients, the gradient
    norm of EVERY parameter and the full gradient of every parameter up to 70 000 elements."""
    from unseenobjectswithmeanshift_amd import training as tr
    from unseenobjectswithmeanshift_amd.modeling import MeanShiftTransformerDecoder
    gd = golden("decoder_backward")
    dec = MeanShiftTransformerDecoder(in_channels=64, mask_classification=True, num_classes=2, hidden_dim=256, num_queries=100, nheads=8,
                                      dim_feedforward=2048, dec_layers=9, pre_norm=False, mask_dim=256, enforce_input_project=False)
    dec.load_state_dict(syn.synth_state_dict(syn.decoder_param_shapes()), strict=True)
    dec = dec.to(DEV)
    x, mf = syn.synth_decoder_inputs(2, 64, 96, seed=1)
    xd = [t.to(DEV).requires_grad_(True) for t in x]
    mfd = mf.to(DEV).requires_grad_(True)
    out = tr.decoder_forward_train(dec, xd, mfd)
    # forward equals the inference path
    with torch.no_grad():
        dec.aux_outputs = True
        inf = dec([t.detach() for t in xd], mfd.detach())
    torch.testing.assert_close(out["pred_masks"].detach(), inf["pred_masks"], rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(out["aux_outputs"][4]["pred_logits"].detach(), inf["aux_outputs"][4]["pred_logits"], rtol=1e-4, atol=1e-4)
    # the same functional as the fixture (weights regenerated from the seed)
    g = torch.Generator().manual_seed(5)
    preds = out["aux_outputs"] + [{"pred_logits": out["pred_logits"], "pred_masks": out["pred_masks"]}]
    loss = 0.0
    for p in preds:
        wl = torch.randn(2, 100, 3, generator=g, dtype=torch.float64)
        wm = torch.randn(2, 100, 16, 24, generator=g, dtype=torch.float64) / (16 * 24) ** 0.5
        loss = loss + (p["pred_logits"] * wl.float().to(DEV)).sum() + (p["pred_masks"] * wm.float().to(DEV)).sum()
    assert abs(float(loss.detach()) - float(gd["loss"])) < 1e-3 * abs(float(gd["loss"])) + 1e-3
    loss.backward()

    def rel(got, ref):
        ref = T(ref).double()
        return float((got.detach().cpu().double() - ref).norm() / ref.norm().clamp_min(1e-12))
    errs = {"x0": rel(xd[0].grad, gd["g_x0"]), "x1": rel(xd[1].grad, gd["g_x1"]), "x2": rel(xd[2].grad, gd["g_x2"]),
            "mf": rel(mfd.grad[:, ::4], gd["g_mf_sub"])}
    print("input gradient relative errors", {k: f"{v:.1e}" for k, v in errs.items()})
    assert max(errs.values()) < 2e-3
    assert abs(float(mfd.grad.norm()) - float(gd["g_mf_norm"])) < 1e-3 * float(gd["g_mf_norm"])
    params = dict(dec.named_parameters())
    worst = (0.0, None)
    for name, ref_norm in zip(gd["param_names"], gd["param_grad_norms"]):
        name = str(name)
        got = params[name].grad
        assert got is not None, name
        assert abs(float(got.norm()) - float(ref_norm)) < 2e-3 * float(ref_norm) + 1e-6, (name, float(got.norm()), float(ref_norm))
        if "g__" + name in gd.files:
            e = rel(got, gd["g__" + name])
            worst = max(worst, (e, name))
            assert e < 3e-3, (name, e)
    print("worst parameter gradient relative error", worst)
