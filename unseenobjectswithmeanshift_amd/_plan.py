"""Execution-plan attributes and the O(1) staleness epoch of captured HIP graphs.

A captured graph (graphs.py) bakes in the kernels a forward launched and the addresses of the derived tensors they read;
it is stale once a module's execution plan changes (precision mode, folded / literal mask step, aux outputs, ...).  The
modules that carry such switches derive from ``PlanAttributes``: assigning one of the names below bumps a process-wide
epoch counter, so a replay compares one integer instead of probing every module for every attribute on every call."""

PLAN_ATTRS = frozenset((
    "precision", "mask_step_dtype", "tails_dtype", "attention_dtype", "kv_split", "sparse_taps", "aux_outputs",
    "folded_mask_features", "batched_kv", "fold_kv", "fused_tails", "fused_encoder", "fused_front", "ffn_parts",
    "fused_kv_attention", "tails_plan", "graphed", "backbone_dtype", "fused_msda", "pooled_attention_masks", "hm_activations",
    "test_topk_per_image", "topk_before_masks"))

_epoch = [0]
_MISSING = object()


def plan_epoch():
    return _epoch[0]


def bump_plan_epoch():
    """Invalidate every captured graph of the process (they re-capture on their next use).  Called automatically when a
    plan attribute of a PlanAttributes module is assigned a new value, and by ``_lib.set_option``."""
    _epoch[0] += 1


class PlanAttributes:
    """Mixin (list it BEFORE nn.Module): assignments to PLAN_ATTRS names bump the plan epoch when the value changes."""

    def __setattr__(self, name, value):
        if name in PLAN_ATTRS:
            old = self.__dict__.get(name, _MISSING)
            if old is _MISSING or old != value:
                bump_plan_epoch()
        super().__setattr__(name, value)


def version_key(tensors):
    """Cheap change detector of a fixed tensor list: (count, sum of version counters, first and last address).  In-place
    updates bump a version, device / dtype moves change the addresses.  ~0.07 us per tensor against ~1 us for a tuple of
    (data_ptr, _version) pairs."""
    if not tensors:
        return (0, 0, 0, 0)
    return (len(tensors), sum([t._version for t in tensors]), tensors[0].data_ptr(), tensors[-1].data_ptr())
