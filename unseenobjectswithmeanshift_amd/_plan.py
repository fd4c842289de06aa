"""Execution-plan attributes and the O(1) staleness epoch of captured HIP graphs.

A captured graph (graphs.py) bakes in the kernels a forward launched and the addresses of the derived tensors they read;
it is stale once a module's execution plan changes (precision mode, folded / literal mask step, aux outputs, ...).  The
modules that carry such switches derive from ``PlanAttributes``: assigning one of the names below bumps a process-wide
epoch counter, so a replay compares one integer instead of probing every module for every attribute on every call."""

PLAN_ATTRS = frozenset((
    "precision", "mask_step_dtype", "tails_dtype", "attention_dtype", "kv_split", "sparse_taps", "aux_outputs",
    "folded_mask_features", "batched_kv", "fold_kv", "fused_tails", "fused_encoder", "fused_front", "ffn_parts",
    "fused_kv_attention", "tails_plan", "graphed", "backbone_dtype", "fused_msda", "pooled_attention_masks", "hm_activations", "lp_input_proj", "lp_prologue", "separable_kv_constants", "lp_operands", "attention_keys", "lp_conv3x3", "fused_kv_min_keys", "gemm_1x1", "fold_mask_conv", "lp_pooled_masks", "fused_epilogues", "miopen_find", "parallel_towers", "fused_head_masks", "weight_prefetch", "tails_hl", "fpn_half_map",
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


# ---- parameter identity --------------------------------------------------------------------------------------------------------
# Derived-tensor caches (packed weight streams, folded constants, captured graphs) are keyed on a cheap signature of the
# parameters they were built from.  The signature must see (a) in-place updates -- every tensor's _version --, (b) storage moves
# (.to(device), .half(), p.data = new) -- every tensor's data_ptr --, and (c) a Parameter OBJECT replaced (m.weight =
# nn.Parameter(...), load_state_dict(assign=True), parametrize): the cached tensor LIST is then stale, so lists are rebuilt when
# the parameter epoch moves.  torch calls the registration hooks below from Module.register_parameter / register_buffer, which
# is where Module.__setattr__ ends for Parameters and buffers.
_param_epoch = [0]


def param_epoch():
    return _param_epoch[0]


def bump_param_epoch(*_args, **_kw):
    _param_epoch[0] += 1


def _install_registration_hooks():
    try:
        from torch.nn.modules import module as _m
        _m.register_module_parameter_registration_hook(bump_param_epoch)
        _m.register_module_buffer_registration_hook(bump_param_epoch)
        return True
    except Exception:                                # an older torch without the global hooks: lists are rebuilt on every key
        return False


_HOOKED = _install_registration_hooks()


class TensorList:
    """The parameters (and buffers) of a module as a list, rebuilt when a Parameter / buffer object was (re)registered anywhere in
    the process since it was built (cheap: module construction is rare on the hot path).

        TensorList.of(module)                              -> module.parameters()
        TensorList.of(module, "transformer.encoder")       -> module.transformer.encoder.parameters()
        TensorList.of(module, buffers=True)                -> parameters + buffers
        TensorList(module.parameters)                      -> a bound method as the builder

    The builder is never a closure over the module: ``copy.deepcopy`` copies functions atomically, so a deep-copied model would
    keep computing its staleness key from the ORIGINAL's tensors (and keep it alive), and ``pickle`` refuses local lambdas.  An
    owner reference / bound method is re-bound to the copy by deepcopy and pickles with the module; the cached list is dropped
    on both."""

    def __init__(self, build=None, owner=None, path="", buffers=False):
        if build is not None and getattr(build, "__self__", None) is None:
            raise TypeError("TensorList: pass a bound method or use TensorList.of(module, path); a plain function / lambda "
                            "would stay bound to the original module under copy.deepcopy")
        self._build = build
        self._owner, self._path, self._buffers = owner, path, bool(buffers)
        self._list = None
        self._epoch = -1

    @classmethod
    def of(cls, owner, path="", buffers=False):
        return cls(None, owner, path, buffers)

    def _tensors(self):
        if self._build is not None:
            return list(self._build())
        m = self._owner
        for name in filter(None, self._path.split(".")):
            m = getattr(m, name)
        out = list(m.parameters())
        if self._buffers:
            out += list(m.buffers())
        return out

    def __call__(self):
        ep = _param_epoch[0]
        if self._list is None or ep != self._epoch or not _HOOKED:
            self._list = self._tensors()
            self._epoch = ep
        return self._list

    def clear(self):
        self._list = None

    def __getstate__(self):
        return {"_build": self._build, "_owner": self._owner, "_path": self._path, "_buffers": self._buffers}

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._list, self._epoch = None, -1

    def __deepcopy__(self, memo):
        import copy
        new = TensorList.__new__(TensorList)
        new.__setstate__({k: copy.deepcopy(v, memo) for k, v in self.__getstate__().items()})
        return new


def version_key(tensors):
    """Cheap change detector of a tensor list: (count, sum of version counters, sum of addresses).  In-place updates bump a
    version; device / dtype moves and ``p.data = new`` on ANY tensor change an address.  Two O(n) integer sums: ~0.2 us per
    tensor against ~1 us for a tuple of (data_ptr, _version) pairs."""
    if not tensors:
        return (0, 0, 0)
    return (len(tensors), sum([t._version for t in tensors]), sum([t.data_ptr() for t in tensors]))


# ---- MIOpen solver search, scoped ----------------------------------------------------------------------------------------------
import contextlib


_find_lock = __import__("threading").RLock()
_find_depth = [0, False]            # nesting depth of active scopes, the flag's value before the outermost one


@contextlib.contextmanager
def miopen_find(on):
    """Inside: torch.backends.cudnn.benchmark = True, i.e. MIOpen measures its solvers for a convolution shape at the first call instead of
    picking one by heuristic ("find" mode).  The flag is process-global: scopes are counted under a lock, the outermost one saves the
    previous value and the last one to leave restores it (two threads running backbones never restore each other's value).  While a HIP
    graph is being captured the flag is left alone -- a solver search must not run inside a capture; the warm-up passes before the
    capture have measured every shape.  (torch.backends.cudnn.flags() would also set the benchmark limit, which MIOpen does not support
    and warns about.)"""
    import torch
    if not on or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        yield
        return
    with _find_lock:
        if _find_depth[0] == 0:
            _find_depth[1] = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = True
        _find_depth[0] += 1
    try:
        yield
    finally:
        with _find_lock:
            _find_depth[0] -= 1
            if _find_depth[0] == 0:
                torch.backends.cudnn.benchmark = _find_depth[1]
