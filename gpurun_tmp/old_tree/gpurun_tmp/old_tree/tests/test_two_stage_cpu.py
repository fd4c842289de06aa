"""Host logic of the two-stage harness (unseenobjectswithmeanshift_amd/two_stage.py, meta_arch.py)
against golden vectors produced by the reference's own functions (lib/fcn/test_utils.py,
lib/fcn/test_dataset.py, lib/fcn/nms.py) on synthetic instances.  CPU only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

from unseenobjectswithmeanshift_amd import two_stage as ts
from unseenobjectswithmeanshift_amd.meta_arch import Instances, combine_masks, get_confident_instances


def harness_inputs(seed, H=96, W=128, n_inst=7):
    """Same recipe as tests/golden/make_golden.py::harness_inputs (kept in sync by the assertions below)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    masks = torch.zeros(n_inst, H, W)
    for i in range(n_inst):
        cy, cx = torch.rand(1, generator=g).item() * H, torch.rand(1, generator=g).item() * W
        ry, rx = 6 + torch.rand(1, generator=g).item() * 18, 6 + torch.rand(1, generator=g).item() * 24
        masks[i] = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1).float()
    scores = torch.rand(n_inst, generator=g) * 0.6 + 0.35
    classes = (torch.rand(n_inst, generator=g) < 0.8).long()
    image = torch.rand(1, 3, H, W, generator=g)
    z = 0.4 + 1.2 * torch.rand(1, 1, H, W, generator=g)
    z[torch.rand(1, 1, H, W, generator=g) < 0.3] = 0
    z[:, :, : H // 3, : W // 3] = 0
    depth = torch.cat([torch.rand(1, 2, H, W, generator=g), z], 1)
    return masks, scores, classes, image, depth


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_harness_against_reference_functions(golden):
    check_harness(golden, "cpu")


def check_harness(golden, dev):
    """The harness functions against the reference's outputs, with every tensor on `dev` (the GPU variant of this test
    lives in test_gpu_modules.py: there the label statistics run in the HIP kernel)."""
    g = golden("harness")
    for case, seed in enumerate((1, 2, 3)):
        masks, scores, classes, image, depth = (t.to(dev) for t in harness_inputs(seed))
        inst = Instances((96, 128), pred_masks=masks, scores=scores, pred_classes=classes)
        conf = get_confident_instances({"instances": inst}, topk=False, score=0.6)
        label = combine_masks(conf)
        assert np.array_equal(label.astype(np.int16), g[f"c{case}_label"])
        conf_topk = get_confident_instances({"instances": inst}, topk=True, low_threshold=0.4)
        assert np.array_equal(combine_masks(conf_topk).astype(np.int16), g[f"c{case}_label_topk"])
        # the sync-free label image used by the pipeline equals the two reference steps
        li = ts.label_image({"instances": inst}, False, 0.6, 0.4, 2)
        assert li.dtype == torch.float64 and np.array_equal(li.cpu().numpy().astype(np.int16), g[f"c{case}_label"])
        li = ts.label_image({"instances": inst}, True, 0.7, 0.4, 2)
        assert np.array_equal(li.cpu().numpy().astype(np.int16), g[f"c{case}_label_topk"])
        bin_mask, score_mask, bbox = ts.combine_masks_with_NMS(conf)
        assert np.array_equal(bin_mask.astype(np.int16), g[f"c{case}_nms_label"])
        assert np.array_equal(score_mask.astype(np.int16), g[f"c{case}_nms_score"])
        assert np.array_equal(bbox, g[f"c{case}_nms_bbox"])
        out_label = torch.as_tensor(label).unsqueeze(0).to(dev)
        filt = ts.filter_labels_depth(out_label, depth, 0.5)
        assert torch.equal(filt.to(torch.int16).cpu(), T(g[f"c{case}_filt"]))
        rgb_crops, mask_crops, rois, depth_crops = ts.crop_rois(image, filt.clone(), depth)
        assert torch.equal(rois.cpu(), T(g[f"c{case}_rois"]))
        tol = 1e-6 if dev == "cpu" else 1e-5            # the GPU's bilinear interpolation orders the four products differently
        torch.testing.assert_close(rgb_crops[:, :, ::3, ::3].cpu(), T(g[f"c{case}_rgb_crops"]), rtol=tol, atol=tol)
        torch.testing.assert_close(depth_crops[:, :, ::3, ::3].cpu(), T(g[f"c{case}_depth_crops"]), rtol=tol, atol=tol)
        bits = np.unpackbits(g[f"c{case}_mask_crops"])[:mask_crops.numel()].reshape(mask_crops.shape)
        assert np.array_equal((mask_crops > 0).cpu().numpy(), bits.astype(bool))
        labels_crop = torch.zeros(rgb_crops.shape[0], 224, 224, device=dev)
        for i in range(rgb_crops.shape[0]):
            labels_crop[i] = mask_crops[i] * (2 + (torch.arange(224, device=dev)[None, :] > 100).float())
            labels_crop[i][:20, :20] = 5
        refined, lc = ts.match_label_crop(filt, labels_crop.clone(), mask_crops, rois, depth_crops)
        assert torch.equal(refined.to(torch.int16).cpu(), T(g[f"c{case}_refined"]))
        assert torch.equal(lc[:, ::2, ::2].to(torch.int8).cpu(), T(g[f"c{case}_labels_crop_out"]))
        refined_nd, _ = ts.match_label_crop(filt, labels_crop.clone(), mask_crops, rois, None)
        assert torch.equal(refined_nd.to(torch.int16).cpu(), T(g[f"c{case}_refined_nodepth"]))


class _FakePredictor:
    """Deterministic stand-in for the network: returns the instances it was built with (first stage)
    or a crop-sized split of whatever it is shown (second stage)."""

    def __init__(self, inst=None):
        self.inst = inst
        self.calls = 0

    def __call__(self, sample):
        self.calls += 1
        if self.inst is not None:
            return {"instances": self.inst}
        h, w = sample["image"].shape[-2:]
        m = torch.zeros(2, h, w)
        m[0, :, : w // 2] = 1
        m[1, :, w // 2:] = 1
        return {"instances": Instances((h, w), pred_masks=m, scores=torch.tensor([0.9, 0.8]), pred_classes=torch.tensor([1, 1]))}

    def batch_call(self, samples):
        return [self(s) for s in samples]


def test_two_stage_pipeline_shapes_and_labels():
    masks, scores, classes, image, depth = harness_inputs(2)
    inst = Instances((96, 128), pred_masks=masks, scores=scores, pred_classes=classes)
    first, second = _FakePredictor(inst), _FakePredictor()
    sample = {"image_color": image[0], "depth": depth[0]}
    out_label, refined, out_score, bbox = ts.test_sample_crop_nolabel(sample, first, second, confident_score=0.6)
    assert out_label.shape == (1, 96, 128) and refined.shape == (1, 96, 128)
    assert out_score is None and bbox is None                     # NMS off: the reference would raise here
    n_rois = int((torch.unique(out_label) != 0).sum())
    assert first.calls == 1 and second.calls == n_rois
    # every refined pixel lies inside a padded ROI of a first-stage object, labels are 1..K contiguous
    ids = torch.unique(refined)
    assert ids[0] == 0 and torch.equal(ids[1:], torch.arange(1, len(ids), dtype=refined.dtype))
    out_label2, refined2, out_score2, bbox2 = ts.test_sample_crop_nolabel(sample, first, None, confident_score=0.6, use_nms=True)
    assert refined2 is None and out_score2.shape == (1, 96, 128) and bbox2.shape[1] == 5


class _BlobPredictor:
    """A deterministic stand-in predictor for host-logic tests: blobs derived from the image content, scores / classes from a seed.
    ``__call__`` is the per-sample interface of the reference's predictor, ``batch_tensors`` what the batched harness prefers."""

    def __init__(self, n_inst=6):
        self.n = n_inst
        self.calls = 0

    def _one(self, sample):
        img = sample["image"]
        H, W = img.shape[-2:]
        seed = int(float(img.sum()) * 1000) % 100003
        g = torch.Generator().manual_seed(seed)
        yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        masks = torch.zeros(self.n, H, W)
        for i in range(self.n):
            cy, cx = torch.rand(1, generator=g).item() * H, torch.rand(1, generator=g).item() * W
            ry, rx = H / 12 + torch.rand(1, generator=g).item() * H / 5, W / 12 + torch.rand(1, generator=g).item() * W / 5
            masks[i] = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1).float()
        scores = torch.rand(self.n, generator=g) * 0.6 + 0.35
        classes = (torch.rand(self.n, generator=g) < 0.8).long()
        return scores, classes, masks

    def __call__(self, sample):
        self.calls += 1
        s, c, m = self._one(sample)
        return {"instances": Instances(tuple(m.shape[-2:]), pred_masks=m, scores=s, pred_classes=c)}

    def batch_call(self, samples):
        return [self(s) for s in samples]

    def batch_tensors(self, samples):
        self.calls += 1
        parts = [self._one(s) for s in samples]
        return tuple(torch.stack([p[i] for p in parts]) for i in range(3))


def test_batched_two_stage_equals_the_frame_by_frame_pipeline():
    """two_stage.test_batch_crop_nolabel (BASELINE configs[3]: a batch of frames end to end) against test_sample_crop_nolabel
    frame by frame -- the reference's own loop structure (lib/fcn/test_utils.py:375-406), whose pieces are pinned above -- with
    the same deterministic predictor in both stages: identical first-stage label images, ROI tables, refined labels.  Host
    logic only (ROI table from one statistics transfer, paste order per frame, per-frame renumbering, crop chunking)."""
    g = torch.Generator().manual_seed(9)
    H, W, Fr = 96, 128, 5
    samples = []
    for f in range(Fr):
        image = torch.rand(3, H, W, generator=g)
        z = 0.4 + 1.2 * torch.rand(1, H, W, generator=g)
        z[torch.rand(1, H, W, generator=g) < 0.3] = 0
        if f == 2:
            z[:] = 0                                     # a frame whose labels are all filtered away: no crops
        depth = torch.cat([torch.rand(2, H, W, generator=g), z], 0)
        samples.append({"image_color": image, "depth": depth, "file_name": "OSD-x" if f == 3 else "f%d" % f})
    kw = dict(topk=False, confident_score=0.5, low_threshold=0.4, num_class=2)
    pred = _BlobPredictor()
    for crop_batch in (256, 4):
        pred.calls = 0
        labels, refined, rows = ts.test_batch_crop_nolabel(samples, pred, pred, use_depth=True, crop_batch=crop_batch, **kw)
        assert labels.shape == (Fr, H, W) and refined.shape == (Fr, H, W)
        assert pred.calls == 1 + -(-len(rows) // crop_batch)
        assert not any(r[0] == 2 for r in rows) and float(refined[2].abs().max()) == 0.0
        for f, smp in enumerate(samples):
            o_label, o_refined, _, _ = ts.test_sample_crop_nolabel(smp, pred, pred, use_depth=True, **kw)
            assert torch.equal(labels[f].double(), o_label[0].double()), f
            if o_refined is None:
                assert float(refined[f].abs().max()) == 0.0
            else:
                assert torch.equal(refined[f].double(), o_refined[0].double()), f
    # without depth: paste order by ROI area; topk selection rule
    labels, refined, rows = ts.test_batch_crop_nolabel(samples, pred, pred, use_depth=False, topk=True, low_threshold=0.5, num_class=2)
    for f, smp in enumerate(samples):
        o_label, o_refined, _, _ = ts.test_sample_crop_nolabel(smp, pred, pred, use_depth=False, topk=True, low_threshold=0.5, num_class=2)
        assert torch.equal(labels[f].double(), o_label[0].double())
        assert torch.equal(refined[f].double(), (o_refined[0] if o_refined is not None else torch.zeros(H, W)).double())
