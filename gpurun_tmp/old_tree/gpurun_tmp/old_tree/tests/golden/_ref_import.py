"""Import the reference hot-path modules in THIS container (CPU) to generate golden vectors.

Only used by ``tests/golden/make_golden.py``.  Never imported by the product, the tests or
the bench: ``/root/reference`` does not exist on the GPU box.

The reference imports detectron2 / fvcore at module top (attention_util.py:2,13-14,
msdeformattn.py:7-15) but the arithmetic of the hot path does not use them, so we register
pass-through stand-ins in ``sys.modules`` for the *import* only (a decorator that returns
the function, a Conv2d that is nn.Conv2d + optional norm/activation, a dict-like Registry).
The package ``__init__`` files (which pull timm / swin) are bypassed by registering bare
namespace modules whose ``__path__`` points at the reference directories.
"""
import importlib
import sys
import types

import torch
from torch import nn

REF_ROOT = "/root/reference"
MSM = REF_ROOT + "/MSMFormer/meanshiftformer"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj


def _configurable(init_func=None, *, from_config=None):
    # pass-through: golden generation always calls constructors with explicit kwargs
    if init_func is not None:
        return init_func
    return lambda f: f


class _Conv2d(nn.Conv2d):
    """nn.Conv2d with detectron2's optional ``norm`` / ``activation`` attributes."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = super().forward(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    assert norm == "GN", norm
    return nn.GroupNorm(32, out_channels)


class _ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


def _c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def _c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


_installed = False


def install_stubs():
    global _installed
    if _installed:
        return
    _installed = True
    _mod("detectron2")
    _mod("detectron2.config", configurable=_configurable)
    _mod("detectron2.layers", Conv2d=_Conv2d, ShapeSpec=_ShapeSpec, get_norm=_get_norm, DeformConv=None)
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=_Registry)
    _mod("detectron2.modeling", SEM_SEG_HEADS_REGISTRY=_Registry("SEM_SEG_HEADS"))
    _mod("fvcore")
    wi = _mod("fvcore.nn.weight_init", c2_xavier_fill=_c2_xavier_fill, c2_msra_fill=_c2_msra_fill)
    _mod("fvcore.nn", weight_init=wi)
    # The native op is absent here; an empty module lets ms_deform_attn_func.py:21-29 import,
    # and the missing attribute routes MSDeformAttn.forward to its own PyTorch fallback
    # (ms_deform_attn.py:116-121).
    _mod("MultiScaleDeformableAttention")
    # namespace packages that skip the reference __init__ files
    for name, path in [
        ("refmsm", MSM),
        ("refmsm.modeling", MSM + "/modeling"),
        ("refmsm.modeling.transformer_decoder", MSM + "/modeling/transformer_decoder"),
        ("refmsm.modeling.pixel_decoder", MSM + "/modeling/pixel_decoder"),
        ("refmsm.modeling.pixel_decoder.ops", MSM + "/modeling/pixel_decoder/ops"),
        ("refmsm.modeling.pixel_decoder.ops.modules", MSM + "/modeling/pixel_decoder/ops/modules"),
        ("refmsm.modeling.pixel_decoder.ops.functions", MSM + "/modeling/pixel_decoder/ops/functions"),
    ]:
        m = _mod(name)
        m.__path__ = [path]
    # the two ops sub-packages re-export through their __init__; mirror that
    f = importlib.import_module("refmsm.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    sys.modules["refmsm.modeling.pixel_decoder.ops.functions"].MSDeformAttnFunction = f.MSDeformAttnFunction
    mm = importlib.import_module("refmsm.modeling.pixel_decoder.ops.modules.ms_deform_attn")
    sys.modules["refmsm.modeling.pixel_decoder.ops.modules"].MSDeformAttn = mm.MSDeformAttn


def ref(name):
    """Import a reference module, e.g. ref('modeling.transformer_decoder.attention_util')."""
    install_stubs()
    return importlib.import_module("refmsm." + name)


def ref_functions(path, names, namespace):
    """Execute ONLY the named top-level function definitions of a reference source file (in memory)
    inside `namespace`.  Used for harness files whose module-level imports need cv2 / easydict /
    detectron2 (lib/fcn/test_dataset.py, lib/fcn/test_utils.py, lib/utils/mask.py, lib/fcn/nms.py):
    the functions themselves only use torch / numpy."""
    import ast
    with open(REF_ROOT + "/" + path) as f:
        tree = ast.parse(f.read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in keep}
    assert not missing, missing
    mod = ast.Module(body=keep, type_ignores=[])
    exec(compile(mod, REF_ROOT + "/" + path, "exec"), namespace)
    return namespace


def ref_method(path, cls_name, names, namespace):
    """Like ref_functions for METHODS: execute only the named ``def``s of class ``cls_name`` of a reference source file as
    plain functions in `namespace` (their first argument stays ``self``: the caller passes a stand-in object)."""
    import ast
    with open(REF_ROOT + "/" + path) as f:
        tree = ast.parse(f.read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name]
    assert cls, cls_name
    keep = [n for n in cls[0].body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in keep}
    assert not missing, missing
    mod = ast.Module(body=keep, type_ignores=[])
    exec(compile(mod, REF_ROOT + "/" + path, "exec"), namespace)
    return namespace
