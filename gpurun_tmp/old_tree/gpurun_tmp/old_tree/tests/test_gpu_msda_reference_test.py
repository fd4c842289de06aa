"""The reference's own test of its native op (OPS/test.py), run through the drop-in module
``MultiScaleDeformableAttention``:

  * check_forward_equal_with_pytorch_double  (OPS/test.py:33-43): float64, torch.allclose default tolerances;
  * check_forward_equal_with_pytorch_float   (OPS/test.py:46-60): float32, rtol 1e-2 / atol 1e-3;
  * check_gradient_numerical                 (OPS/test.py:66-89): torch.autograd.gradcheck in float64 for
    D in {30, 32, 64, 71, 1025, 2048, 3096} (OPS/test.py:84-85).

The forward checks run on the reference test's OWN inputs and against the reference's OWN outputs: tests/golden/msda_core.npz holds
the tensors that test draws (seed 3, its sizes: N, M, D = 1, 2, 2; Lq, L, P = 2, 2, 2; levels (6, 4), (3, 2)) and what
``ms_deform_attn_core_pytorch`` returned for them when tests/golden/make_golden.py imported the reference -- nothing of the
reference's test is restated here.  gradcheck needs no particular inputs (it compares the op with its own finite differences):
they are drawn locally.  The op is reached through ``sys.modules["MultiScaleDeformableAttention"]``, the name the reference's
function file imports, and through the package's autograd wrapper.  /root/reference is not read."""
import os
import sys

import numpy as np
import pytest
import torch
from torch.autograd import gradcheck

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msda_core.npz")


def _setup():
    import unseenobjectswithmeanshift_amd.MultiScaleDeformableAttention as shim
    sys.modules["MultiScaleDeformableAttention"] = shim
    import MultiScaleDeformableAttention as MSDA          # the import statement the reference's function file uses
    from unseenobjectswithmeanshift_amd.training import MSDeformAttnFunction
    assert MSDA is shim and callable(MSDA.ms_deform_attn_forward) and callable(MSDA.ms_deform_attn_backward)
    shapes = torch.tensor([(6, 4), (3, 2)], dtype=torch.long, device="cuda")
    start = torch.tensor([0, 24], dtype=torch.long, device="cuda")
    return MSDeformAttnFunction, shapes, start


def _inputs(channels, seed):
    """Generic op inputs at the small geometry: locations inside and slightly outside [0, 1], positive weights."""
    g = torch.Generator().manual_seed(seed)
    value = torch.randn(1, 30, 2, channels, generator=g, dtype=torch.float64) * 0.1
    loc = torch.rand(1, 2, 2, 2, 2, 2, generator=g, dtype=torch.float64) * 1.1 - 0.05
    aw = torch.softmax(torch.randn(1, 2, 2, 4, generator=g, dtype=torch.float64), -1).view(1, 2, 2, 2, 2)
    return value.cuda(), loc.cuda(), aw.cuda()


def test_reference_op_test_forward_double_then_float():
    """OPS/test.py:33-60 on the tensors that test draws, against the outputs the reference's PyTorch op gave for them."""
    from oracle import msm_oracle as O
    Fn, shapes, start = _setup()
    g = np.load(GOLDEN)
    T = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        value, loc, aw, ref = T("t_double_value").double(), T("t_double_loc").double(), T("t_double_aw").double(), T("t_double_out")
        assert ref.dtype == torch.float64
        out = Fn.apply(value.cuda(), shapes, start, loc.cuda(), aw.cuda(), 2).cpu()
        assert out.dtype == torch.float64
        assert torch.allclose(out, ref)                                          # the reference's check: default rtol 1e-5, atol 1e-8
        assert ((out - ref).abs() / ref.abs()).max() < 1e-12                     # it is float64 arithmetic, not a cast
        assert torch.allclose(O.ms_deform_attn_core_grid_sample(value, shapes.cpu(), loc, aw), ref)          # (the oracle agrees)
        value, loc, aw, ref = T("t_float_value"), T("t_float_loc"), T("t_float_aw"), T("t_float_out")
        out = Fn.apply(value.cuda(), shapes, start, loc.cuda(), aw.cuda(), 2).cpu()
        assert out.dtype == torch.float32
        assert torch.allclose(out, ref, rtol=1e-2, atol=1e-3)                    # the reference's check
        assert torch.allclose(out, ref, rtol=1e-5, atol=1e-8)                    # and far inside it


@pytest.mark.parametrize("channels", [30, 32, 64, 71, 1025, 2048, 3096])
def test_reference_op_test_gradcheck(channels):
    """OPS/test.py:66-89 with the channel counts of :84-85: numerical vs analytical Jacobian of the float64 op with
    respect to value, sampling locations and attention weights (gradcheck defaults: eps 1e-6, atol 1e-5, rtol 1e-3,
    nondet_tol 0 -- the analytical pass must also be bit-reproducible)."""
    Fn, shapes, start = _setup()
    value, loc, aw = _inputs(channels, seed=3 + channels)
    value.requires_grad = True
    loc.requires_grad = True
    aw.requires_grad = True
    assert gradcheck(Fn.apply, (value, shapes, start, loc, aw, 2))


def test_float_kernels_beyond_64_channels():
    """fp32 with D > 64 (outside the tuned kernels) takes the shape-generic kernel: forward and backward agree with
    the float64 instantiation to fp32 rounding."""
    from unseenobjectswithmeanshift_amd import ops
    _, shapes, start = _setup()
    value, loc, aw = (t.float() for t in _inputs(71, seed=11))
    go = torch.rand(1, 2, 2 * 71, generator=torch.Generator().manual_seed(12)).cuda()
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
