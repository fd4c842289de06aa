"""Load the reference's published checkpoints into this package's modules.

The reference saves detectron2 checkpoints (``torch.save({"model": state_dict, ...})``, README.md:86-95: MSMFormer weights
for the ResNet-50 and the UCN configurations, plus the separate UCN ``seg_resnet34_8s_embedding`` checkpoints that the
meta-arch loads by itself, pretrained_meanshiftformer_model.py:29-70).  The hot-path modules here keep the reference's
parameter names, so conversion is a matter of prefixes:

    pretrained_backbone.*            -> backbone.*          PretrainedMeanShiftMaskFormer names its backbone that way
                                                            (pretrained_meanshiftformer_model.py:148-158), this package's
                                                            meta-archs call it ``backbone``
    backbone.*                       -> backbone.*          MeanShiftMaskFormer (meanshiftformer_model.py)
    sem_seg_head.pixel_decoder.*     -> unchanged
    sem_seg_head.predictor.*         -> unchanged           (``static_query`` -> ``query_feat`` of v1 checkpoints is migrated by
                                                            the decoder's own _load_from_state_dict, as in the reference)
    criterion.*, pixel_mean, pixel_std -> dropped (training-only / non-persistent); *.num_batches_tracked -> dropped unless
                                          the model keeps such a buffer (the UCN towers do)
    module.* (DistributedDataParallel wrapper)                 -> stripped

UCN ``SEGNET`` checkpoints (lib/networks/SEG.py) carry ``fcn.*`` / ``fcn_depth.*`` (optionally under ``module.``): they load
into ``UCNBackbone`` directly (``convert_ucn_state_dict``).
"""
import torch

_DROP_PREFIXES = ("criterion.",)
_DROP_KEYS = ("pixel_mean", "pixel_std")


def _unwrap(obj):
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        obj = obj["model"]
    if isinstance(obj, dict) and "state_dict" in obj and isinstance(obj["state_dict"], dict):
        obj = obj["state_dict"]
    return obj


def convert_reference_state_dict(state_dict):
    """Reference meta-arch checkpoint -> the key layout of meta_arch.MeanShiftMaskFormer / PretrainedMeanShiftMaskFormer.
    Accepts the raw ``torch.load`` result (with or without the {"model": ...} wrapper).  Values are converted to fp32 tensors
    (detectron2 stores numpy arrays in converted model-zoo pickles)."""
    out = {}
    for k, v in _unwrap(state_dict).items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.startswith(_DROP_PREFIXES) or k in _DROP_KEYS:
            continue
        if k.startswith("pretrained_backbone."):
            k = "backbone." + k[len("pretrained_backbone."):]
        t = torch.as_tensor(v)
        out[k] = t.float() if t.is_floating_point() else t
    return out


def convert_ucn_state_dict(state_dict):
    """UCN SEGNET checkpoint (``fcn.*`` / ``fcn_depth.*``) -> ucn_backbone.UCNBackbone keys."""
    out = {}
    for k, v in _unwrap(state_dict).items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.endswith("num_batches_tracked") or not k.startswith(("fcn.", "fcn_depth.")):
            continue
        out[k] = torch.as_tensor(v).float()
    return out


def load_checkpoint_file(path, unsafe=False):
    """torch.load of a checkpoint file, tensors-only by default: the published checkpoints are downloaded from third-party
    links (README.md:86-95) and a full unpickle executes whatever the file says.  detectron2-style checkpoints hold
    tensors, numpy arrays and plain containers, so the numpy reconstructors are allow-listed; ``unsafe=True`` is the
    explicit opt-in to a full unpickle for legacy files that hold other objects."""
    if unsafe:
        return torch.load(path, map_location="cpu", weights_only=False)
    import pickle

    import numpy as np
    allow = [np.ndarray, np.dtype]
    core = getattr(np, "_core", None) or getattr(np, "core")
    for name in ("_reconstruct", "scalar"):
        fn = getattr(core.multiarray, name, None)
        if fn is not None:
            allow.append(fn)
            # torch matches allow-listed globals by "module.name": a file pickled under numpy 1.x names numpy.core.multiarray.*,
            # one pickled under numpy 2.x numpy._core.multiarray.* -- register the reconstructors under BOTH paths
            for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):
                allow.append((fn, f"{mod}.{name}"))
    allow += [type(np.dtype(t)) for t in ("float32", "float64", "float16", "int64", "int32", "uint8", "bool")]
    def _load(globals_):
        with torch.serialization.safe_globals(globals_):
            return torch.load(path, map_location="cpu", weights_only=True)

    plain = [a for a in allow if not isinstance(a, tuple)]
    try:
        try:
            return _load(allow)
        except (TypeError, AttributeError):            # a torch whose allow-list does not take (callable, "module.name") pairs: it accepts the
            return _load(plain)                        # tuple into the list and fails inside torch.load ('tuple' has no __module__)
    except pickle.UnpicklingError as e:
        raise pickle.UnpicklingError(
            f"{path}: not loadable tensors-only ({e}).  Legacy detectron2 checkpoints may hold other objects (trainer state, "
            "numpy scalars pickled by another numpy major version): if you trust the file, load it with unsafe=True "
            "(load_checkpoint_file / load_reference_checkpoint), or re-save its 'model' entry as plain tensors") from e


def load_reference_checkpoint(model, checkpoint, strict=True, unsafe=False):
    """``checkpoint``: a path (``torch.load``-able) or an already loaded object.  ``model``: a meta-arch of this package.
    With ``strict`` every parameter / buffer of the model must be present and nothing may be left over; a model built
    without a backbone (features handed over by the caller) ignores the checkpoint's backbone.*.  Files are read
    tensors-only (``load_checkpoint_file``) unless ``unsafe=True``.  Returns the converted state dict."""
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        checkpoint = load_checkpoint_file(checkpoint, unsafe=unsafe)
    sd = convert_reference_state_dict(checkpoint)
    if getattr(model, "backbone", None) is None:
        sd = {k: v for k, v in sd.items() if not k.startswith("backbone.")}
    # BatchNorm step counters: kept where the model has them (the UCN towers keep torchvision's BatchNorm layout), dropped where
    # it does not (a backbone whose frozen BatchNorm is folded away)
    have = set(model.state_dict())
    sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked") or k in have}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if strict and (missing or unexpected):
        raise RuntimeError(f"load_reference_checkpoint: missing keys {sorted(missing)[:8]}{'...' if len(missing) > 8 else ''}, "
                           f"unexpected keys {sorted(unexpected)[:8]}{'...' if len(unexpected) > 8 else ''}")
    return sd
