"""Head, meta-architecture and predictor around the hot path.

  MeanShiftMaskFormerHead   <- modeling/meta_arch/meanshift_former_head.py:18-143
  MeanShiftMaskFormer       <- meanshiftformer_model.py / pretrained_meanshiftformer_model.py:244-378
                               (eval branch) and instance_inference :461-497
  Network_RGBD              <- lib/fcn/test_utils.py:150-166 (predictor __call__(sample) -> dict)
  Instances                 <- the three detectron2 containers the harness reads, reduced to a
                               field bag (pred_masks / pred_boxes / scores / pred_classes)

detectron2 is not a dependency.  The backbone is out of scope: the
meta-arch takes any ``backbone`` module mapping an image batch to {"res2".."res5"}; ``None`` means
the caller already passes backbone features.  Inference only.
"""
import inspect

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from ._plan import PlanAttributes

_ACCEPTS = {}


def _accepts(method, name):
    """Does ``method`` take a parameter called ``name``?  (cached per function: inspect.signature costs ~15 us a call)"""
    f = getattr(method, "__func__", method)
    key = (f, name)
    if key not in _ACCEPTS:
        _ACCEPTS[key] = name in inspect.signature(f).parameters
    return _ACCEPTS[key]


class Instances:
    """Minimal stand-in for detectron2.structures.Instances: attribute bag + boolean/index slicing
    (what get_confident_instances / combine_masks use, lib/fcn/test_utils.py:35-112)."""

    def __init__(self, image_size, **fields):
        self.image_size = tuple(image_size)
        self._fields = dict(fields)

    def __getattr__(self, name):
        f = self.__dict__.get("_fields", {})
        if name in f:
            return f[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in ("image_size", "_fields"):
            super().__setattr__(name, value)
        else:
            self._fields[name] = value

    def get(self, name):
        return self._fields[name]

    def has(self, name):
        return name in self._fields

    def get_fields(self):
        return self._fields

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        return 0

    def __getitem__(self, item):
        if isinstance(item, torch.Tensor) and item.dtype == torch.bool:
            item = item.to(next(iter(self._fields.values())).device)
        return Instances(self.image_size, **{k: v[item] for k, v in self._fields.items()})

    def to(self, device):
        return Instances(self.image_size, **{k: v.to(device) for k, v in self._fields.items()})


class MeanShiftMaskFormerHead(PlanAttributes, nn.Module):
    _version = 2

    def __init__(self, input_shape, *, num_classes, pixel_decoder, loss_weight=1.0, ignore_value=-1,
                 transformer_predictor, transformer_in_feature="multi_scale_pixel_decoder"):
        super().__init__()
        if transformer_in_feature != "multi_scale_pixel_decoder":
            raise NotImplementedError("only TRANSFORMER_IN_FEATURE == 'multi_scale_pixel_decoder'")
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.ignore_value = ignore_value
        self.common_stride = 4
        self.loss_weight = loss_weight
        self.pixel_decoder = pixel_decoder
        self.predictor = transformer_predictor
        self.transformer_in_feature = transformer_in_feature
        self.num_classes = num_classes

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # head-prefix migration of old checkpoints (meanshift_former_head.py:23-45)
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for k in list(state_dict.keys()):
                if k.startswith(prefix) and "sem_seg_head" in k and not k.startswith(prefix + "predictor") \
                        and not k.startswith(prefix + "pixel_decoder"):
                    state_dict[k.replace(prefix, prefix + "pixel_decoder.")] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def forward(self, features, image_height=None, image_width=None, mask=None, final_topk=0):
        return self.layers(features, image_height, image_width, mask, final_topk)

    def set_precision(self, mode):
        """"f32" (the reference's arithmetic, default), "bf16" (BASELINE configs 3 / 5): bf16 MFMA operands with fp32
        accumulation in the encoder's token-wise GEMMs, the decoder's row-local tails, the attention cores and the
        Q x pixel-embedding mask step; everything that decides a sign or normalises (LayerNorms, softmax, unit-norm, the
        residual streams) stays fp32 -- "f16": the same 16-bit plan with IEEE-half operands (v_mfma_f32_16x16x32_f16, the bf16
        instruction's rate) wherever the operand's range is bounded -- Linear weights and LayerNorm'd activations of the tails and of
        the encoder's FFN, unit-norm keys -- and bf16 where it is not (softmax weights, value rows): 2^-12 instead of 2^-9 roundings
        on the products that carry the plan's error -- or "f32_split": fp32 everywhere, the encoder's GEMMs, the K/V projection and the
        (folded) mask step as exact three-term bf16 splits on the bf16 matrix pipe (fp32-accurate, see csrc/enc_block_split.hip)."""
        if mode not in ("f32", "f32_split", "bf16", "f16"):
            raise ValueError("precision must be 'f32', 'f32_split', 'bf16' or 'f16'")
        self.precision = mode
        lowp = mode in ("bf16", "f16")
        if hasattr(self.pixel_decoder, "precision"):
            self.pixel_decoder.precision = "bf16" if lowp else mode          # the low-precision PLAN; its operand format below
        if hasattr(self.pixel_decoder, "lp_operands"):
            self.pixel_decoder.lp_operands = "f16" if mode == "f16" else "bf16"
        low = "bf16" if lowp else "f32"
        self.predictor.mask_step_dtype = "f32_split" if mode == "f32_split" else (mode if lowp else "f32")
        if hasattr(self.predictor, "tails_dtype"):
            self.predictor.tails_dtype = mode if lowp else "f32"
        if hasattr(self.predictor, "attention_dtype"):
            self.predictor.attention_dtype = low
        if hasattr(self.predictor, "attention_keys"):
            self.predictor.attention_keys = "f16" if mode == "f16" else "bf16"
        if hasattr(self.predictor, "kv_split"):
            self.predictor.kv_split = mode == "f32_split"
        return self

    def layers(self, features, image_height=None, image_width=None, mask=None, final_topk=0):
        """Returns (predictions, last_feature_map).  The reference also upsamples mask_features to
        image size here (meanshift_former_head.py:121-126, 315 MB per 640x480 image) for the
        training-only embedding loss; inference returns None for it."""
        # a predictor that contracts mask_features in their factored form gets them that way (modeling.FoldedMaskFeatures):
        # same predictions up to fp32 summation order, a quarter of the mask step's work
        kw = {}
        if getattr(self.predictor, "folded_mask_features", False) and _accepts(self.pixel_decoder.forward_features, "folded"):
            kw["folded"] = True
        mask_features, _, multi_scale_features = self.pixel_decoder.forward_features(features, **kw)
        if final_topk and _accepts(self.predictor.forward, "final_topk"):
            predictions = self.predictor(multi_scale_features, mask_features, mask, final_topk=final_topk)
        else:
            predictions = self.predictor(multi_scale_features, mask_features, mask)
        return predictions, None


class PretrainedMeanShiftMaskFormerHead(MeanShiftMaskFormerHead):
    """meanshift_former_head.py:145-275: identical glue for the UCN (RGB-D) configuration; the reference
    calls exit() for any TRANSFORMER_IN_FEATURE other than "multi_scale_pixel_decoder" (:258-274)."""


class MeanShiftMaskFormer(PlanAttributes, nn.Module):
    """Inference branch of the meta-arch (pretrained_meanshiftformer_model.py:244-303,334-378; meanshiftformer_model.py:
    214-245,286-330).

    Input normalisation: the reference has two meta-archs.  ``MeanShiftMaskFormer`` (meanshiftformer_model.py:115-116,241)
    owns non-persistent ``pixel_mean`` / ``pixel_std`` buffers and computes (x - mean) / std BEFORE padding to the size
    divisibility, so the padded border is zero in normalised space; pass ``pixel_mean`` / ``pixel_std`` (cfg.MODEL.PIXEL_MEAN /
    PIXEL_STD, Base-COCO-InstanceSegmentation.yaml:6-7) to get that.  ``PretrainedMeanShiftMaskFormer`` -- the meta-arch
    every shipped yaml selects, also for the ResNet-50 backbone (mixture_ResNet50.yaml:27, USE_OTHER_BACKBONE) -- does NOT
    normalise (pretrained_meanshiftformer_model.py:270-275: the dataset / predictor hands over normalised images,
    lib/datasets/ocid_dataset.py:105-106); that is the default here (``pixel_mean=None``)."""

    def __init__(self, *, backbone, sem_seg_head, num_queries, test_topk_per_image=20, size_divisibility=32,
                 instance_on=True, pixel_mean=None, pixel_std=None):
        super().__init__()
        if (pixel_mean is None) != (pixel_std is None):
            raise ValueError("pixel_mean and pixel_std go together")
        if pixel_mean is not None:
            self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
            self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        else:
            self.pixel_mean = self.pixel_std = None
        self.backbone = backbone
        self.sem_seg_head = sem_seg_head
        self.num_queries = num_queries
        self.test_topk_per_image = test_topk_per_image
        self.size_divisibility = size_divisibility
        self.instance_on = instance_on

    @torch.no_grad()
    def inference(self, features, image_size, padded_size=None):
        """features: dict res2..res5 (B,C,h,w) on the GPU.  Returns the per-batch tensors
        (scores (B,T), classes (B,T), masks (B,T,H,W), boxes (B,T,4), query_index (B,T)).  ``padded_size``: the frame
        the features were computed on when the image was padded to the size divisibility (masks are cropped back to
        image_size, PM:275,354-357)."""
        padded_size = tuple(padded_size or image_size)
        # the final mask step only for the queries kept below (an argument, not module state: two pipelines may share one model)
        k = int(self.test_topk_per_image) if getattr(self, "topk_before_masks", True) else 0
        if k and _accepts(self.sem_seg_head.forward, "final_topk"):
            outputs, _ = self.sem_seg_head(features, padded_size[0], padded_size[1], final_topk=k)
        else:
            outputs, _ = self.sem_seg_head(features, padded_size[0], padded_size[1])
        if "topk" in outputs:
            cls_scores, classes, qidx = outputs["topk"]
            B, K = qidx.shape
            key = (B, K, str(qidx.device))
            if getattr(self, "_iota", (None,))[0] != key:
                self._iota = (key, torch.arange(K, device=qidx.device, dtype=torch.int32)[None].expand(B, -1).contiguous())
            local = self._iota[1]                                  # pred_masks holds exactly the K selected queries, in order
        else:
            cls_scores, classes, qidx = ops.topk_class_scores(outputs["pred_logits"], self.test_topk_per_image)
            local = qidx
        # scores = class prob * mean mask prob (PM:495), fused into the post-process kernel
        masks, scores, boxes = ops.instance_postprocess(outputs["pred_masks"], local, image_size, class_scores=cls_scores,
                                                        padded_size=padded_size)
        return scores, classes, masks, boxes, qidx

    @torch.no_grad()
    def inference_images(self, inputs, image_size, padded_size=None):
        """``inference`` with the backbone in front: inputs {"image": (B,3,Hp,Wp)[, "depth": (B,3,Hp,Wp)]} already padded to the
        size divisibility (and normalised, if this meta-arch normalises).  One call = the whole model; ``graphed(entry=
        "inference_images")`` replays it from a HIP graph (MIOpen's convolutions capture like any other launch once their
        algorithms have been chosen by the warm-up passes)."""
        if self.backbone is None:
            raise RuntimeError("inference_images needs a backbone")
        feats = self.backbone(inputs["image"], inputs["depth"]) if "depth" in inputs else self.backbone(inputs["image"])
        return self.inference(feats, image_size, padded_size)

    def set_precision(self, mode):
        """See MeanShiftMaskFormerHead.set_precision; a backbone with a ``backbone_dtype`` switch (ResNet50Backbone) follows."""
        self.sem_seg_head.set_precision(mode)
        if hasattr(self.backbone, "backbone_dtype"):
            self.backbone.backbone_dtype = mode if mode in ("bf16", "f16") else "f32"     # ("f16": IEEE-half convolutions, as the reference's autocast)
        return self

    @property
    def precision(self):
        return getattr(self.sem_seg_head, "precision", "f32")

    def graphed(self, warmup=2, entry="inference"):
        """HIP-graph replayed ``inference`` (graphs.GraphedInference): same results, no per-launch host cost.
        entry="inference_images": the backbone is part of the graph (inputs {"image": ...})."""
        from .graphs import GraphedInference
        return GraphedInference(self, warmup=warmup, entry=entry)

    def pipelined(self, depth=2, warmup=2, entry="inference"):
        """Throughput mode (graphs.PipelinedInference): ``depth`` batches in flight, one HIP graph and stream each.
        entry="inference_images": the backbone is part of every slot's graph (inputs {"image": ...[, "depth": ...]})."""
        from .graphs import PipelinedInference
        return PipelinedInference(self, depth=depth, warmup=warmup, entry=entry)

    @torch.no_grad()
    def forward(self, batched_inputs):
        """batched_inputs: list of dicts with "image" (3,H,W) -- or one dict holding a 4-D batch, as
        the reference accepts (PM:270-273) -- plus, when ``backbone`` is None, "features"."""
        first = batched_inputs[0]
        div = self.size_divisibility
        if self.backbone is None:
            feats = first["features"] if isinstance(first["features"], dict) and first["features"]["res2"].dim() == 4 \
                else {k: torch.stack([x["features"][k] for x in batched_inputs]) for k in first["features"]}
            padded = (4 * feats["res2"].shape[-2], 4 * feats["res2"].shape[-1])       # the frame the features cover
            H, W = first.get("height"), first.get("width")
            if H is None:
                H, W = padded
            if not (padded[0] - div < H <= padded[0] and padded[1] - div < W <= padded[1]):
                raise ValueError(f"height/width {H}x{W} do not fit features of a {padded[0]}x{padded[1]} frame")
        else:
            images = first["image"] if first["image"].dim() == 4 else torch.stack([x["image"] for x in batched_inputs])
            H, W = images.shape[-2:]
            if first.get("height", H) != H or first.get("width", W) != W:
                raise NotImplementedError("output height/width other than the image size (sem_seg_postprocess resize, PM:354)")
            padded = (-(-H // div) * div, -(-W // div) * div)
            if self.pixel_mean is not None:                                   # meanshiftformer_model.py:241, before the padding
                images = (images - self.pixel_mean) / self.pixel_std
            if padded != (H, W):            # ImageList.from_tensors(images, size_divisibility): zeros at the right / bottom
                images = F.pad(images, (0, padded[1] - W, 0, padded[0] - H))
            feats = self.backbone(images)
        scores, classes, masks, boxes, _ = self.inference(feats, (int(H), int(W)), padded)
        results = []
        for b in range(scores.shape[0]):
            inst = Instances((int(H), int(W)), pred_masks=masks[b], pred_boxes=boxes[b], scores=scores[b],
                             pred_classes=classes[b])
            results.append({"instances": inst})
        return results


class PretrainedMeanShiftMaskFormer(MeanShiftMaskFormer):
    """The RGB-D (UCN backbone) meta-arch, eval branch (pretrained_meanshiftformer_model.py:280-301): the pretrained
    embedding network sees the image and the xyz depth map, its L2-normalised 64-channel full-resolution output is the
    single feature level 'res5' of the head.  ``backbone(img, label, depth)`` follows SEGNET.forward (ucn_backbone.py)."""

    def __init__(self, *, backbone, sem_seg_head, num_queries, use_depth=True, **kw):
        super().__init__(backbone=backbone, sem_seg_head=sem_seg_head, num_queries=num_queries, **kw)
        self.use_depth = use_depth

    @torch.no_grad()
    def forward(self, batched_inputs):
        first = batched_inputs[0]
        images = first["image"] if first["image"].dim() == 4 else torch.stack([x["image"] for x in batched_inputs])
        depth = None
        if self.use_depth:
            depth = first["depth"] if first["depth"].dim() == 4 else torch.stack([x["depth"] for x in batched_inputs])
        H, W = int(images.shape[-2]), int(images.shape[-1])
        if first.get("height", H) != H or first.get("width", W) != W:
            raise NotImplementedError("output height/width other than the image size (sem_seg_postprocess resize, PM:354)")
        div = max(int(self.size_divisibility), 1)
        padded = (-(-H // div) * div, -(-W // div) * div)
        if padded != (H, W):                      # ImageList.from_tensors: zeros at the right / bottom (PM:275, 286)
            images = F.pad(images, (0, padded[1] - W, 0, padded[0] - H))
            depth = None if depth is None else F.pad(depth, (0, padded[1] - W, 0, padded[0] - H))
        scores, classes, masks, boxes, _ = self.inference_images({"image": images, **({} if depth is None else {"depth": depth})}, (H, W), padded)
        return [{"instances": Instances((H, W), pred_masks=masks[b], pred_boxes=boxes[b], scores=scores[b], pred_classes=classes[b])}
                for b in range(scores.shape[0])]

    def inference_images(self, inputs, image_size, padded_size=None):
        """The whole RGB-D model on padded inputs {"image": (B,3,Hp,Wp)[, "depth": xyz (B,3,Hp,Wp)]}: the two towers (SEG.py:88-117,
        ``backbone(img, label=None, depth)``), the channel normalisation of pretrained_meanshiftformer_model.py:298-300, the head
        and the post-processing; ``graphed(entry="inference_images")`` / ``pipelined`` replay exactly this."""
        if self.backbone is None:
            raise RuntimeError("inference_images needs a backbone")
        depth = inputs.get("depth") if self.use_depth else None
        if _accepts(self.backbone.forward, "renormalize"):
            # PM:298-300 (F.normalize over channels) inside the backbone's fused tail: no further pass over the embedding
            feats = {"res5": self.backbone(inputs["image"], None, depth, renormalize=True).float().contiguous()}
            return self.inference(feats, image_size, padded_size)
        feats = self.backbone(inputs["image"], None, depth)
        feats = feats.float().contiguous()
        if feats.is_cuda:
            feats = {"res5": ops.l2_normalize_nchw(feats)}                                # PM:298-300 (F.normalize over channels)
        else:
            feats = {"res5": F.normalize(feats, p=2, dim=1).contiguous()}
        return self.inference(feats, image_size, padded_size)


def build_ucn_model(num_queries=100, dec_layers=6, use_depth=True, **head_kw):
    """mixture_UCN.yaml end to end: UCN ResNet34-8s RGB-D backbone -> SimpleBasePixelDecoder -> 6-layer hypersphere
    decoder over every pixel -> top-k instance post-processing (random-init; load checkpoints with load_state_dict)."""
    from .ucn_backbone import UCNBackbone
    head = build_ucn_head(num_queries=num_queries, dec_layers=dec_layers, **head_kw)
    return PretrainedMeanShiftMaskFormer(backbone=UCNBackbone(num_units=64, in_channels=3, use_depth=use_depth), sem_seg_head=head,
                                         num_queries=num_queries, use_depth=use_depth)


class Network_RGBD:
    """lib/fcn/test_utils.py:150-166: ``predictor(sample) -> {"instances": ...}`` for one sample."""

    def __init__(self, model):
        self.model = model.eval()

    def __call__(self, sample):
        with torch.no_grad():
            return self.model([sample])[0]


# ----------------------------------------------------------------------------------------------
# harness helpers (host side), lib/fcn/test_utils.py:35-112
# ----------------------------------------------------------------------------------------------
def get_confident_instances(outputs, topk=False, score=0.7, num_class=2, low_threshold=0.4):
    instances = outputs["instances"]
    if topk:
        if num_class >= 2:
            instances = instances[instances.pred_classes == 1]
            return instances[instances.scores > low_threshold]
        return instances
    return instances[instances.scores > score]


def combine_masks(instances):
    """(N,H,W) 0/1 masks -> (H,W) label image, labels 2..N+1, later instances overwrite earlier ones
    (test_utils.py:93-112).  Returns a float64 numpy array like the reference."""
    import numpy as np
    mask = instances.get("pred_masks").to("cpu").numpy()
    num, h, w = mask.shape if mask.ndim == 3 else (0, *instances.image_size)
    out = np.zeros((h, w))
    for m, lab in zip(mask, range(2, 2 + len(mask))):
        out[np.nonzero(m)] = lab
    return out


def combine_masks_tensor(instances):
    """combine_masks without leaving the device: (H,W) float64 tensor, identical values.  "Later instances overwrite
    earlier ones" with labels growing in instance order is the per-pixel maximum of mask_i * (i + 2)."""
    masks = instances.get("pred_masks")
    if masks.dim() != 3 or masks.shape[0] == 0:
        h, w = instances.image_size
        return torch.zeros((h, w), dtype=torch.float64, device=masks.device)
    ids = torch.arange(2, 2 + masks.shape[0], device=masks.device, dtype=torch.float64)
    return ((masks != 0).to(torch.float64) * ids[:, None, None]).amax(0)


# ----------------------------------------------------------------------------------------------
def build_ucn_head(num_queries=100, dec_layers=6, num_classes=2, hidden_dim=256, mask_dim=256, conv_dim=64, nheads=8,
                   dim_feedforward=2048):
    """The configuration of MSMFormer/configs/mixture_UCN.yaml:40-66 (RGB-D path): SimpleBasePixelDecoder +
    PretrainedMeanShiftTransformerDecoder over the full-resolution 64-channel embedding ("res5")."""
    from .modeling import PretrainedMeanShiftTransformerDecoder, ShapeSpec, SimpleBasePixelDecoder
    shape = {"res5": ShapeSpec(channels=conv_dim, stride=1)}
    pd = SimpleBasePixelDecoder(shape, conv_dim=conv_dim, mask_dim=mask_dim, norm="GN")
    dec = PretrainedMeanShiftTransformerDecoder(in_channels=conv_dim, mask_classification=True, num_classes=num_classes,
                                                hidden_dim=hidden_dim, num_queries=num_queries, nheads=nheads,
                                                dim_feedforward=dim_feedforward, dec_layers=dec_layers, pre_norm=False,
                                                mask_dim=mask_dim, enforce_input_project=False)
    return PretrainedMeanShiftMaskFormerHead(shape, num_classes=num_classes, pixel_decoder=pd, transformer_predictor=dec,
                                             transformer_in_feature="multi_scale_pixel_decoder")


def build_resnet50_model(num_queries=100, dec_layers=9, **head_kw):
    """mixture_ResNet50.yaml end to end: detectron2-layout ResNet-50 (resnet_backbone.ResNet50Backbone) -> MSDeformAttn pixel
    decoder -> 9-layer hypersphere decoder -> top-k instance post-processing, under the meta-arch every shipped yaml selects
    (PretrainedMeanShiftMaskFormer with USE_OTHER_BACKBONE: no pixel normalisation inside the model).  Random-init; load the
    published weights with checkpoint.load_reference_checkpoint(model, path)."""
    from .resnet_backbone import ResNet50Backbone
    head = build_resnet50_head(num_queries=num_queries, dec_layers=dec_layers, **head_kw)
    return MeanShiftMaskFormer(backbone=ResNet50Backbone(), sem_seg_head=head, num_queries=num_queries)


def build_resnet50_head(num_queries=100, dec_layers=9, num_classes=2, hidden_dim=256, mask_dim=256, conv_dim=64,
                        nheads=8, dim_feedforward=2048, enc_layers=6):
    """The configuration of MSMFormer/configs/mixture_ResNet50.yaml:31-77 (hot path only)."""
    from .modeling import MeanShiftTransformerDecoder, MSDeformAttnPixelDecoder, ShapeSpec
    shape = {"res2": ShapeSpec(channels=256, stride=4), "res3": ShapeSpec(channels=512, stride=8),
             "res4": ShapeSpec(channels=1024, stride=16), "res5": ShapeSpec(channels=2048, stride=32)}
    pd = MSDeformAttnPixelDecoder(shape, transformer_dropout=0.0, transformer_nheads=nheads,
                                  transformer_dim_feedforward=1024, transformer_enc_layers=enc_layers,
                                  conv_dim=conv_dim, mask_dim=mask_dim, norm="GN",
                                  transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    dec = MeanShiftTransformerDecoder(in_channels=conv_dim, mask_classification=True, num_classes=num_classes,
                                      hidden_dim=hidden_dim, num_queries=num_queries, nheads=nheads,
                                      dim_feedforward=dim_feedforward, dec_layers=dec_layers, pre_norm=False,
                                      mask_dim=mask_dim, enforce_input_project=False,
                                      use_meanshift_cross_attention=True, disable_attention_mask=False,
                                      use_meanshift_self_attention=True, decoder_block_norm=True)
    return MeanShiftMaskFormerHead(shape, num_classes=num_classes, pixel_decoder=pd, transformer_predictor=dec,
                                   transformer_in_feature="multi_scale_pixel_decoder")
