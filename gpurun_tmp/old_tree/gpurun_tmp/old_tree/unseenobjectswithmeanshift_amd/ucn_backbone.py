"""UCN RGB-D embedding backbone (inference): two dilated ResNet34-8s towers, add fusion, unit-norm 64-d embedding.

SURVEY.md section 8 f rank 4 ("next" row): the network in front of the PretrainedMeanShiftTransformerDecoder / of the
classic mean-shift clustering.  Reference: lib/networks/SEG.py:24-117 (SEGNET with network_name
'seg_resnet34_8s_embedding': `fcn` on the image, `fcn_depth` on the xyz depth map, FUSION_TYPE 'add',
EMBEDDING_NORMALIZATION), lib/networks/resnet_dilated.py:287-327 (Resnet34_8s: torchvision-style ResNet34 with the
strides of layer3 / layer4 replaced by dilations 2 / 4, the classifier turned into a 1x1 convolution to the embedding
width, bilinear upsampling with align_corners=True back to the input size), lib/networks/resnet.py:43-73,150-258.

These are stock 3x3 convolutions: they run through torch's convolution (MIOpen) -- no hand-written kernel is
warranted.  What is done for the MI355X: inference folds every BatchNorm into its convolution once per checkpoint
(a conv + bias + ReLU chain with no separate normalisation pass over the 1/8-resolution maps), both towers run in
channels_last, and the module keeps the reference's parameter names, so `SEGNET` checkpoints load unchanged
(`fcn.resnet34_8s.layer1.0.conv1.weight`, ..., `fcn_depth.resnet34_8s.fc.bias`).
"""
import torch
import torch.nn.functional as F
from torch import nn

from ._plan import PlanAttributes, miopen_find

STAGES = ((64, 3, 1, 1), (128, 4, 2, 1), (256, 6, 1, 2), (512, 3, 1, 4))   # (planes, blocks, stride, dilation) at output stride 8


class _Block(nn.Module):
    """BasicBlock (resnet.py:43-73): conv3x3-BN-ReLU-conv3x3-BN, + shortcut, ReLU."""

    def __init__(self, cin, planes, stride, dilation, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes)) if project else None


class _DilatedResNet34(nn.Module):
    def __init__(self, num_units, in_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (planes, blocks, stride, dil) in enumerate(STAGES):
            # resnet.py:203-246: once the output stride (8) is reached a stage's stride becomes a dilation; the first
            # block of a stage has a projection shortcut when the width (or the nominal stride) changes
            layer = [_Block(cin, planes, stride, dil, project=(i > 0))]
            layer += [_Block(planes, planes, 1, dil, project=False) for _ in range(blocks - 1)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*layer))
            cin = planes
        self.fc = nn.Conv2d(512, num_units, 1)


class _Tower(nn.Module):
    """Resnet34_8s (resnet_dilated.py:287-327)."""

    def __init__(self, num_units, in_channels):
        super().__init__()
        self.resnet34_8s = _DilatedResNet34(num_units, in_channels)


def _fold(conv, bn):
    """conv (no bias) followed by an eval-mode BatchNorm == conv with scaled weights and a bias."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return (conv.weight * scale[:, None, None, None]).contiguous(memory_format=torch.channels_last), bn.bias - bn.running_mean * scale


_TOWER_STREAMS = {}


def _tower_stream(device):
    key = str(device)
    if key not in _TOWER_STREAMS:
        _TOWER_STREAMS[key] = torch.cuda.Stream(device=device)
    return _TOWER_STREAMS[key]


class UCNBackbone(PlanAttributes, nn.Module):
    """``forward(img, label=None, depth=None) -> (B, num_units, H, W)`` unit-norm embedding, as SEGNET.forward for
    INPUT 'RGBD' / FUSION_TYPE 'add' (SEG.py:88-117); with ``depth=None`` only the colour tower runs (INPUT 'COLOR')."""

    def __init__(self, num_units=64, in_channels=3, use_depth=True, normalize=True):
        super().__init__()
        self.fcn = _Tower(num_units, in_channels)
        self.fcn_depth = _Tower(num_units, in_channels) if use_depth else None
        self.normalize = normalize
        # "bf16" / "f16" (MeanShiftMaskFormer.set_precision("bf16") / ("f16")): the towers' convolutions run in bfloat16 / IEEE half through
        # MIOpen (fp32 accumulation inside the library), the fusion add, the upsampling and the normalisation in fp32.  Half is what the
        # reference's own low-precision mode runs convolutions in (torch.autocast on CUDA defaults to float16) and is the faster of the two here:
        # MIOpen's bf16 solvers accumulate into an fp32 workspace they zero and cast around every convolution (4.1 against 6.4 ms for both
        # towers at batch 2); activations beyond the half range saturate at 65504 in the fused epilogues
        self.backbone_dtype = "f32"
        self.fused_epilogues = True        # bias + ReLU / bias + residual + ReLU around the library convolutions as one HIP launch each
        self.miopen_find = True            # MIOpen measures its solvers per convolution shape at the first call (see forward)
        self.parallel_towers = True        # the depth tower on a second stream beside the colour tower (see _forward)
        self._folded = None
        self._lp = None

    def _plan(self):
        """Per tower: the folded (weight, bias, stride, padding, dilation) of every convolution, rebuilt when a parameter
        or a BatchNorm buffer changes."""
        towers = [self.fcn] + ([self.fcn_depth] if self.fcn_depth is not None else [])
        key = tuple((t.data_ptr(), t._version) for tw in towers for t in list(tw.parameters()) + list(tw.buffers()))
        if self._folded is None or self._folded[0] != key:
            plans = []
            with torch.no_grad():
                for tw in towers:
                    net = tw.resnet34_8s
                    stem = _fold(net.conv1, net.bn1)
                    blocks = []
                    for i in range(4):
                        for blk in getattr(net, f"layer{i + 1}"):
                            c1, c2 = _fold(blk.conv1, blk.bn1), _fold(blk.conv2, blk.bn2)
                            sc = _fold(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None
                            blocks.append((c1, c2, sc, blk.conv1.stride, blk.conv1.dilation))
                    plans.append((stem, blocks, net.fc.weight, net.fc.bias))
            self._folded = (key, plans)
        return self._folded[1]

    def _run(self, plan, x, upsample=True):
        (w, b), blocks, fcw, fcb = plan
        size = x.shape[2:]
        # the elementwise glue of a BasicBlock (resnet_dilated.py / torchvision BasicBlock.forward: bias + ReLU, bias + residual + ReLU) as one
        # HIP launch each instead of the bias kernel MIOpen appends + F.relu + add + F.relu (csrc/backbone_ops.hip)
        fuse = getattr(self, "fused_epilogues", True) and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16)
        if fuse:
            from . import ops
        cl = lambda t: t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)

        def conv_act(t, wt, bs, res=None, **kw):
            if not fuse:
                y = F.conv2d(t, wt, bs, **kw)
                return F.relu(y if res is None else y + res)
            return ops.bias_act_nhwc_(cl(F.conv2d(t, wt, None, **kw)), bs.contiguous(), None if res is None else cl(res), True)

        x = conv_act(x.contiguous(memory_format=torch.channels_last), w, b, stride=2, padding=3)
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        for (w1, b1), (w2, b2), sc, stride, dil in blocks:
            y = conv_act(x, w1, b1, stride=stride, padding=dil, dilation=dil)
            if sc is not None:
                x = F.conv2d(x, sc[0], sc[1], stride=stride)
            x = conv_act(y, w2, b2, res=x, padding=dil, dilation=dil)
        x = F.conv2d(x, fcw, fcb).float()
        if not upsample:
            return x                                                                 # (the fused tail upsamples both towers at once)
        return F.interpolate(x, size=size, mode="bilinear", align_corners=True)      # nn.functional.upsample_bilinear

    @torch.no_grad()
    def forward(self, img, label=None, depth=None, *, renormalize=False):
        """``renormalize`` (not a reference argument): apply the channel normalisation once more, as the meta-arch does to the
        backbone's output (pretrained_meanshiftformer_model.py:298-300) -- inside the fused tail instead of in another pass."""
        # MIOpen "find" mode for this module's convolutions (solvers measured once per shape at the first call): towers 5.6 -> 3.5 ms in bf16
        with miopen_find(bool(getattr(self, "miopen_find", True)) and img.is_cuda):
            return self._forward(img, label, depth, renormalize)

    def _forward(self, img, label, depth, renormalize):
        if self.training:
            raise NotImplementedError("UCNBackbone is an inference module (BatchNorm folded into the convolutions): call .eval()")
        plans = self._plan()
        if self.backbone_dtype not in ("f32", "bf16", "f16"):
            raise ValueError("backbone_dtype must be 'f32', 'bf16' or 'f16'")
        dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[self.backbone_dtype]
        if dt != torch.float32:                   # a 16-bit copy of the folded weights, made once per parameter version
            if self._lp is None or self._lp[0] is not self._folded or self._lp[2] != dt:
                c = lambda t: None if t is None else t.to(dt)
                self._lp = (self._folded, [((c(w), c(b)), [((c(w1), c(b1)), (c(w2), c(b2)), None if sc is None else (c(sc[0]), c(sc[1])), st, dl)
                                                          for (w1, b1), (w2, b2), sc, st, dl in blocks], c(fcw), c(fcb)) for (w, b), blocks, fcw, fcb in plans], dt)
            plans = self._lp[1]
        if depth is not None and self.fcn_depth is None:
            raise RuntimeError("this backbone was built without a depth tower")
        if self.fused_epilogues and img.is_cuda and plans[0][2].shape[0] == 64:
            # upsampling of both towers, add fusion and the normalisation(s) in ONE pass over the output (csrc/backbone_ops.hip,
            # ucn_tail_kernel) instead of eight passes of torch ops over the full-resolution embedding
            from . import ops
            if depth is not None and getattr(self, "parallel_towers", True):
                # the colour and the depth tower are independent (SEG.py:97-110): the depth tower runs on a second stream -- its 1/8-resolution
                # convolutions are a few hundred tiles each and leave most of the chip idle on their own (a HIP-graph capture records the
                # fork and the join as a parallel branch)
                cur = torch.cuda.current_stream()
                side = _tower_stream(img.device)
                xd = depth.float().to(dt)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    lo_b = self._run(plans[1], xd, upsample=False)
                lo_a = self._run(plans[0], img.float().to(dt), upsample=False)
                cur.wait_stream(side)
                lo_b.record_stream(cur)
                xd.record_stream(side)
            else:
                lo_a = self._run(plans[0], img.float().to(dt), upsample=False)
                lo_b = self._run(plans[1], depth.float().to(dt), upsample=False) if depth is not None else None
            return ops.ucn_embedding_tail(lo_a, lo_b, img.shape[2:], norms=(1 if self.normalize else 0) + (1 if renormalize else 0))
        feats = self._run(plans[0], img.float().to(dt))
        if depth is not None:
            feats = feats + self._run(plans[1], depth.float().to(dt))
        if self.normalize:
            feats = F.normalize(feats, p=2, dim=1)
        if renormalize:
            feats = F.normalize(feats, p=2, dim=1)
        return feats.contiguous()
