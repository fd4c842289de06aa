"""Drop-in for the reference's pybind module ``MultiScaleDeformableAttention``
(ops/src/vision.cpp:18-21): same two function names and positional signatures, so the reference's
``MSDeformAttnFunction`` (ops/functions/ms_deform_attn_func.py:32-49) works unmodified once this
module is importable under that name:

    import sys, unseenobjectswithmeanshift_amd.MultiScaleDeformableAttention as m
    sys.modules["MultiScaleDeformableAttention"] = m

float32 and float64 like the reference's dispatch (ms_deform_attn_cuda.cu:69,139): the reference's own op test
(ops/test.py: double forward check, float forward check, double gradcheck for D up to 3096) runs through this module as
written (tests/test_gpu_msda_reference_test.py).
"""
import torch

from . import ops


def _check_inputs(value, tensors, im2col_step):
    step = min(value.shape[0], int(im2col_step))
    if value.shape[0] % step != 0:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (value.shape[0], step))
    for t, name in tensors:
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")      # cu:33-37 / cu:98-103
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")             # cu:39-43 / cu:105-110
    if value.dtype not in (torch.float32, torch.float64):                   # AT_DISPATCH_FLOATING_TYPES, cu:69 / cu:139
        raise RuntimeError(f"ms_deform_attn is implemented for float and double, got {value.dtype}")


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """ops/src/ms_deform_attn.h:25-44.  The batch-chunking argument only has to satisfy the
    reference's divisibility check (ms_deform_attn_cuda.cu:55-57); one launch covers the batch."""
    _check_inputs(value, ((value, "value"), (spatial_shapes, "spatial_shapes"), (level_start_index, "level_start_index"),
                          (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight")), im2col_step)
    return ops.ms_deform_attn(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    """ops/src/ms_deform_attn.h:46-66 -> [grad_value, grad_sampling_loc, grad_attn_weight]
    (ms_deform_attn_cuda.cu:88-158), as MSDeformAttnFunction.backward expects (ms_deform_attn_func.py:41-49)."""
    _check_inputs(value, ((value, "value"), (spatial_shapes, "spatial_shapes"), (level_start_index, "level_start_index"),
                          (sampling_loc, "sampling_loc"), (attn_weight, "attn_weight"), (grad_output, "grad_output")),
                  im2col_step)
    gv, gl, gw = ops.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                             grad_output)
    return [gv, gl, gw]
