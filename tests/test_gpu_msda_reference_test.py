"""The reference's own test of its native op (OPS/test.py), run through the drop-in module
``MultiScaleDeformableAttention`` exactly as that test drives the CUDA extension:

  * check_forward_equal_with_pytorch_double  (OPS/test.py:33-43): float64, torch.allclose default tolerances;
  * check_forward_equal_with_pytorch_float   (OPS/test.py:46-60): float32, rtol 1e-2 / atol 1e-3;
  * check_gradient_numerical                 (OPS/test.py:66-89): torch.autograd.gradcheck in float64 for
    D in {30, 32, 64, 71, 1025, 2048, 3096} (OPS/test.py:84-85).

Same sizes (N, M, D = 1, 2, 2; Lq, L, P = 2, 2, 2; levels (6, 4), (3, 2)), same seed (3), same order of random draws, the
op reached through ``sys.modules["MultiScaleDeformableAttention"]`` with the reference's autograd-Function pattern
(OPS/functions/ms_deform_attn_func.py:32-49).  The comparison partner is the oracle's grid_sample form of the op
(ms_deform_attn_func.py:52-72), evaluated on the CPU.  /root/reference is not read."""
import sys

import pytest
import torch
from torch.autograd import gradcheck

pytestmark = pytest.mark.gpu

N, M, D = 1, 2, 2
Lq, L, P = 2, 2, 2


def _setup():
    import unseenobjectswithmeanshift_amd.MultiScaleDeformableAttention as shim
    sys.modules["MultiScaleDeformableAttention"] = shim
    import MultiScaleDeformableAttention as MSDA          # the import statement the reference's function file uses

    class MSDeformAttnFunction(torch.autograd.Function):
        @staticmethod
        def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
            ctx.im2col_step = im2col_step
            output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                                 attention_weights, ctx.im2col_step)
            ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
            return output

        @staticmethod
        def backward(ctx, grad_output):
            value, shapes, start, loc, aw = ctx.saved_tensors
            gv, gl, gw = MSDA.ms_deform_attn_backward(value, shapes, start, loc, aw, grad_output.contiguous(), ctx.im2col_step)
            return gv, None, None, gl, gw, None

    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    level_start_index = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    return MSDeformAttnFunction, shapes, level_start_index, S


def _draw(S, channels):
    value = torch.rand(N, S, M, channels).cuda() * 0.01
    sampling_locations = torch.rand(N, Lq, M, L, P, 2).cuda()
    attention_weights = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
    attention_weights /= attention_weights.sum(-1, keepdim=True).sum(-2, keepdim=True)
    return value, sampling_locations, attention_weights


def test_reference_op_test_forward_double_then_float():
    """OPS/test.py:33-60, in the order __main__ runs them (one seed, consecutive draws)."""
    from oracle import msm_oracle as O
    Fn, shapes, start, S = _setup()
    torch.manual_seed(3)
    with torch.no_grad():
        value, loc, aw = _draw(S, D)
        ref = O.ms_deform_attn_core_grid_sample(value.double().cpu(), shapes.cpu(), loc.double().cpu(), aw.double().cpu())
        out = Fn.apply(value.double(), shapes, start, loc.double(), aw.double(), 2).detach().cpu()
        assert out.dtype == torch.float64
        assert torch.allclose(out, ref), (out - ref).abs().max()                 # default rtol 1e-5, atol 1e-8
        rel = ((out - ref).abs() / ref.abs()).max()
        assert rel < 1e-12                                                       # it is float64 arithmetic, not a cast
        value, loc, aw = _draw(S, D)
        ref = O.ms_deform_attn_core_grid_sample(value.cpu(), shapes.cpu(), loc.cpu(), aw.cpu())
        out = Fn.apply(value, shapes, start, loc, aw, 2).detach().cpu()
        assert out.dtype == torch.float32
        assert torch.allclose(out, ref, rtol=1e-2, atol=1e-3)
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-8)                    # and far inside the reference's own bound


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025, 2048, 3096])
def test_reference_op_test_gradcheck(channels):
    """OPS/test.py:66-89 with the channel counts of :84-85: numerical vs analytical Jacobian of the float64 op with
    respect to value, sampling locations and attention weights (gradcheck defaults: eps 1e-6, atol 1e-5, rtol 1e-3,
    nondet_tol 0 -- the analytical pass must also be bit-reproducible)."""
    Fn, shapes, start, S = _setup()
    torch.manual_seed(3 + channels)
    value, loc, aw = _draw(S, channels)
    value.requires_grad = True
    loc.requires_grad = True
    aw.requires_grad = True
    assert gradcheck(Fn.apply, (value.double(), shapes, start, loc.double(), aw.double(), 2))


def test_float_kernels_beyond_64_channels():
    """fp32 with D > 64 (outside the tuned kernels) takes the shape-generic kernel: forward and backward agree with
    the float64 instantiation to fp32 rounding."""
    from unseenobjectswithmeanshift_amd import ops
    Fn, shapes, start, S = _setup()
    torch.manual_seed(11)
    value, loc, aw = _draw(S, 71)
    go = torch.rand(N, Lq, M * 71).cuda()
    o32 = ops.ms_deform_attn(value, shapes, start, loc, aw)
    o64 = ops.ms_deform_attn(value.double(), shapes, start, loc.double(), aw.double())
    assert torch.allclose(o32.double(), o64, rtol=1e-5, atol=1e-8)
    g32 = ops.ms_deform_attn_backward(value, shapes, start, loc, aw, go)
    g64 = ops.ms_deform_attn_backward(value.double(), shapes, start, loc.double(), aw.double(), go.double())
    for a, b in zip(g32, g64):
        assert torch.allclose(a.double(), b, rtol=1e-4, atol=1e-6 * float(b.abs().max()) + 1e-9)
    with pytest.raises(RuntimeError):
        ops.ms_deform_attn(value.half(), shapes, start, loc.half(), aw.half())
    with pytest.raises(RuntimeError):
        ops.ms_deform_attn(value.double(), shapes, start, loc, aw)               # mixed types
