"""detectron2-layout ResNet-50 backbone (inference) producing res2..res5 for the MSDeformAttn pixel decoder.

SURVEY.md section 8 f rank 4 ("next" row).  The shipped ResNet-50 configuration (MSMFormer/configs/mixture_ResNet50.yaml:23
USE_OTHER_BACKBONE, Base-COCO-InstanceSegmentation.yaml:2-15) builds detectron2's ``build_resnet_backbone`` with DEPTH 50,
STEM_OUT_CHANNELS 64, STRIDE_IN_1X1 False, OUT_FEATURES res2..res5 and the default FrozenBN norm, and the meta-arch calls it
as ``self.pretrained_backbone(images.tensor)`` (pretrained_meanshiftformer_model.py:277-279).

detectron2 is not importable in this environment, so this module is written from the architecture those config keys
select -- stem: 7x7/2 convolution (pad 3) + FrozenBN + ReLU + 3x3/2 max pool (pad 1); res2..res5: 3, 4, 6, 3 bottleneck blocks
(1x1 -> 3x3 -> 1x1, widths 64/128/256/512 -> 256/512/1024/2048, ReLU after the residual add), the stride 2 of res3..res5 on
the 3x3 convolution of the first block (STRIDE_IN_1X1 False), a 1x1 projection shortcut with the same stride on every first
block -- and keeps detectron2's parameter names so that the published checkpoints load unchanged:
``stem.conv1.weight``, ``stem.conv1.norm.{weight,bias,running_mean,running_var}``, ``res2.0.shortcut.weight``,
``res2.0.conv1.weight``, ``res2.0.conv1.norm.weight`` ... ``res5.2.conv3.norm.running_var``.  PARITY UNPINNED: there is no
reference implementation here to generate golden vectors from; tests check the structure (keys, shapes, strides), the
BatchNorm folding against the unfolded definition in float64, and -- on the GPU -- the folded fp32 network against the same
float64 evaluation.

These are stock convolutions: they run through torch's convolution (MIOpen).  What is done for the MI355X: every frozen
BatchNorm is folded into its convolution once per checkpoint (conv + bias + ReLU chains, no normalisation passes), the
network runs in channels_last, and the four outputs are returned as contiguous NCHW fp32 maps -- the layout the pixel
decoder's input projections stream.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ._plan import PlanAttributes, TensorList, miopen_find, version_key

STAGE_BLOCKS = {"res2": 3, "res3": 4, "res4": 6, "res5": 3}          # DEPTH 50
STAGE_WIDTHS = {"res2": (64, 256), "res3": (128, 512), "res4": (256, 1024), "res5": (512, 2048)}     # (bottleneck, out)
STAGE_STRIDE = {"res2": 1, "res3": 2, "res4": 2, "res5": 2}
OUT_STRIDES = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine parameters, all four as buffers (detectron2.layers.FrozenBatchNorm2d):
    y = (x - running_mean) / sqrt(running_var + eps) * weight + bias."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.view(1, -1, 1, 1).to(x.dtype) + shift.view(1, -1, 1, 1).to(x.dtype)


class _ConvBN(nn.Conv2d):
    """detectron2.layers.Conv2d(..., bias=False, norm=FrozenBN): parameters .weight and .norm.*"""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.norm = FrozenBatchNorm2d(cout)

    def forward(self, x):
        return self.norm(super().forward(x))

    def folded(self):
        scale = self.norm.weight * (self.norm.running_var + self.norm.eps).rsqrt()
        return ((self.weight * scale[:, None, None, None]).contiguous(memory_format=torch.channels_last),
                (self.norm.bias - self.norm.running_mean * scale).contiguous())


class BasicStem(nn.Module):
    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.conv1 = _ConvBN(in_channels, out_channels, 7, stride=2, padding=3)

    def forward(self, x):
        return F.max_pool2d(F.relu(self.conv1(x)), 3, stride=2, padding=1)


class BottleneckBlock(nn.Module):
    def __init__(self, cin, cout, bottleneck, stride):
        super().__init__()
        self.shortcut = _ConvBN(cin, cout, 1, stride=stride) if cin != cout else None
        self.conv1 = _ConvBN(cin, bottleneck, 1)                                  # STRIDE_IN_1X1 False: the stride sits on conv2
        self.conv2 = _ConvBN(bottleneck, bottleneck, 3, stride=stride, padding=1)
        self.conv3 = _ConvBN(bottleneck, cout, 1)

    def forward(self, x):
        y = self.conv3(F.relu(self.conv2(F.relu(self.conv1(x)))))
        return F.relu(y + (x if self.shortcut is None else self.shortcut(x)))


class ResNet50Backbone(PlanAttributes, nn.Module):
    """``forward(images (B,3,H,W)) -> {"res2": (B,256,H/4,W/4), "res3": (B,512,H/8,W/8), "res4": (B,1024,H/16,W/16),
    "res5": (B,2048,H/32,W/32)}``; H, W multiples of 32 (the meta-arch pads).  ``folded=False`` evaluates the unfolded
    definition (conv, frozen BN, ReLU as separate ops) -- the reference the folding is tested against."""

    def __init__(self, in_channels=3, out_features=("res2", "res3", "res4", "res5")):
        super().__init__()
        self.stem = BasicStem(in_channels, 64)
        cin = 64
        for name in ("res2", "res3", "res4", "res5"):
            bott, cout = STAGE_WIDTHS[name]
            blocks = [BottleneckBlock(cin, cout, bott, STAGE_STRIDE[name])]
            blocks += [BottleneckBlock(cout, cout, bott, 1) for _ in range(STAGE_BLOCKS[name] - 1)]
            setattr(self, name, nn.Sequential(*blocks))
            cin = cout
        self.out_features = tuple(out_features)
        self.size_divisibility = 32
        self._plan_cache = None
        self._plan_tensors = None
        # "f32": MIOpen fp32 convolutions (default).  "bf16": the folded weights and the activations in bfloat16 (MIOpen's
        # bf16 convolutions accumulate in fp32), the four output maps converted back to fp32 -- the low-precision mode of
        # BASELINE configs[2] / [4] (the reference's counterpart is autocast over the whole model)
        self.backbone_dtype = "f32"
        # the 1x1 convolutions (two thirds of the network's FLOPs) as plain GEMMs on the NHWC view of their channels_last input --
        # hipBLASLt through torch.addmm, bias (+ ReLU) in its epilogue -- instead of MIOpen's convolution: batch 8 at 480x640 on one
        # MI355X 7.9 -> 5.8 ms in fp32, 5.9 -> 2.95 ms in bf16 (tools/probes/resnet_gemm_time.py); same arithmetic, another summation order
        self.gemm_1x1 = True
        # the elementwise glue around the convolutions (bias + ReLU, bias + residual + ReLU, the NCHW fp32 hand-over) as one HIP launch each
        # instead of the bias kernel MIOpen appends, F.relu, the residual add and the conversions: see csrc/backbone_ops.hip
        self.fused_epilogues = True
        self.miopen_find = True            # let MIOpen measure its solvers per convolution shape at the first call (see forward)

    def output_shape(self):
        from .modeling import ShapeSpec
        return {k: ShapeSpec(channels=STAGE_WIDTHS[k][1], stride=OUT_STRIDES[k]) for k in self.out_features}

    def _plan(self):
        if self._plan_tensors is None:
            self._plan_tensors = TensorList.of(self, buffers=True)
        if self.backbone_dtype not in ("f32", "bf16", "f16"):
            raise ValueError("backbone_dtype must be 'f32', 'bf16' or 'f16'")
        low = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}[self.backbone_dtype]
        key = version_key(self._plan_tensors()) + (low,)
        if self._plan_cache is None or self._plan_cache[0] != key:
            with torch.no_grad():
                stages = []
                for name in ("res2", "res3", "res4", "res5"):
                    stages.append([(blk.conv1.folded(), blk.conv2.folded(), blk.conv3.folded(),
                                    None if blk.shortcut is None else blk.shortcut.folded(), blk.conv2.stride) for blk in getattr(self, name)])
                stem = self.stem.conv1.folded()
                if low is not None:
                    cast = lambda wb: None if wb is None else (wb[0].to(low).contiguous(memory_format=torch.channels_last), wb[1].to(low))
                    stem = cast(stem)
                    stages = [[(cast(a), cast(b), cast(c), cast(sc), st) for a, b, c, sc, st in blocks] for blocks in stages]
                self._plan_cache = (key, stem, stages)
        return self._plan_cache[1:]

    @torch.no_grad()
    def forward(self, images, folded=True):
        # MIOpen picks a convolution's solver by heuristic unless asked to measure ("find" mode = torch.backends.cudnn.benchmark): measured once
        # per shape at the first call (seconds), it is 2.70 -> 1.95 ms in bf16, 5.3 -> 4.8 ms in fp32 at batch 8.  Scoped to this module's calls.
        with miopen_find(bool(self.miopen_find) and images.is_cuda):
            return self._forward(images, folded)

    def _forward(self, images, folded=True):
        if self.training:
            raise NotImplementedError("ResNet50Backbone is an inference module (frozen BatchNorm folded into the convolutions): call .eval()")
        out = {}
        if not folded:
            x = self.stem(images)
            for name in ("res2", "res3", "res4", "res5"):
                x = getattr(self, name)(x)
                if name in self.out_features:
                    out[name] = x
            return out
        (ws, bs), stages = self._plan()
        x = images.to(ws.dtype).contiguous(memory_format=torch.channels_last)
        gemm = self.gemm_1x1 and x.is_cuda
        # the elementwise glue around the library convolutions in one launch each (csrc/backbone_ops.hip): a convolution's bias + ReLU,
        # a block's bias + residual add + ReLU, the NHWC -> NCHW fp32 hand-over to the pixel decoder
        fuse = self.fused_epilogues and x.is_cuda and ws.dtype in (torch.float32, torch.bfloat16, torch.float16)
        if fuse:
            from . import ops
        cl = lambda t: t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)

        def conv_bias_relu(t, w, b, **kw):
            if not fuse:
                return F.relu(F.conv2d(t, w, b, **kw))
            return ops.bias_act_nhwc_(cl(F.conv2d(t, w, None, **kw)), b, None, True)

        x = F.max_pool2d(conv_bias_relu(x, ws, bs, stride=2, padding=3), 3, stride=2, padding=1)

        def conv1x1(t, wb, relu, stride=1):
            w, b = wb
            if not gemm:
                y = F.conv2d(t, w, b, stride=stride)
                return F.relu(y) if relu else y
            if stride != 1:
                t = t[:, :, ::stride, ::stride].contiguous(memory_format=torch.channels_last)
            B, C, H, W = t.shape
            a = t.permute(0, 2, 3, 1).reshape(B * H * W, C)                 # the NHWC view of a channels_last map: no copy
            w2d = w.flatten(1)                                              # (Cout, Cin, 1, 1) -> (Cout, Cin)
            y = torch._addmm_activation(b, a, w2d.t(), use_gelu=False) if relu else torch.addmm(b, a, w2d.t())
            return y.view(B, H, W, -1).permute(0, 3, 1, 2)                  # (B, Cout, H, W) in channels_last memory

        for name, blocks in zip(("res2", "res3", "res4", "res5"), stages):
            for c1, (w2, b2), c3, sc, stride in blocks:
                st = stride[0] if isinstance(stride, tuple) else stride
                y = conv1x1(x, c1, True)
                y = conv_bias_relu(y, w2, b2, stride=stride, padding=1)
                res = x if sc is None else conv1x1(x, sc, False, st)
                if fuse and gemm:
                    B_, _, H_, W_ = y.shape
                    y3 = torch.mm(y.permute(0, 2, 3, 1).reshape(B_ * H_ * W_, -1), c3[0].flatten(1).t()).view(B_, H_, W_, -1).permute(0, 3, 1, 2)
                    x = ops.bias_act_nhwc_(y3, c3[1], cl(res), True)              # conv3's bias, the residual and the block's ReLU in one pass
                else:
                    x = F.relu(conv1x1(y, c3, False) + res)
            if name in self.out_features:
                # NCHW planes for the pixel decoder's input projections (fp32 also in the bf16 mode)
                out[name] = ops.nhwc_to_nchw_f32(cl(x)) if fuse else (x.float() if x.dtype in (torch.bfloat16, torch.float16) else x).contiguous()
        return out
