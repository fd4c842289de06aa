"""Two-stage (zoom-in refinement) harness around the predictor.

Own counterpart of the reference's host-side harness, same function names / argument meaning /
return values:

  filter_labels_depth       <- lib/fcn/test_dataset.py:183-198
  crop_rois                 <- lib/fcn/test_dataset.py:62-112
  match_label_crop          <- lib/fcn/test_dataset.py:116-179
  nms                       <- lib/fcn/nms.py:3-23
  combine_masks_with_NMS    <- lib/fcn/test_utils.py:55-91
  test_sample_crop_nolabel  <- lib/fcn/test_utils.py:339-421

These are data-dependent, tiny (<= 20 instances) bookkeeping steps; they run as torch ops on
whatever device the label maps live on (GPU in production, CPU in the unit tests), never through
the oracle.  Differences from the reference, on purpose:
  * the second stage is BATCHED: all crops of an image go through the crop predictor in one call
    (the reference loops batch-1, test_utils.py:396-405);
  * test_sample_crop_nolabel returns (out_label, out_label_refined, out_score, bbox) with None for
    the last two when NMS is off -- the reference raises NameError there (test_utils.py:376,421).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .meta_arch import combine_masks_tensor, combine_masks, get_confident_instances

CROP_SIZE = 224          # cfg.TRAIN.SYN_CROP_SIZE, lib/fcn/config.py:130
PADDING_PERCENTAGE = 0.25


LABEL_BINS = 4096      # label images hold 0 and 2..N+1 with N <= detections per image (20 by default)


def mask_to_tight_box(mask):
    """lib/utils/mask.py:179-186: (x_min, y_min, x_max, y_max) of the non-zero pixels."""
    ys, xs = torch.nonzero(mask, as_tuple=True)
    return xs.min(), ys.min(), xs.max(), ys.max()


def filter_labels_depth(labels, depth, threshold):
    """Zero every label whose pixels have valid depth (z > 0) on less than `threshold` of their area.
    labels (B,H,W) with small non-negative integer values, depth (B,3,H,W) xyz.  (lib/fcn/test_dataset.py:183-198; the
    per-label loop of the reference is two histograms here: same integer counts, same fp32 division, no host syncs.)"""
    out = labels.clone()
    for i in range(labels.shape[0]):
        lab = labels[i].reshape(-1).to(torch.int64)
        k = int(LABEL_BINS)
        valid = (depth[i, 2] > 0).reshape(-1)
        area = torch.bincount(lab.clamp(0, k - 1), minlength=k)[:k]
        good = torch.bincount(lab.clamp(0, k - 1)[valid], minlength=k)[:k]
        bad = (good.float() / area.float().clamp_min(1.0) < threshold) & (area > 0)
        bad[0] = False
        out[i][bad[lab.clamp(0, k - 1)].view_as(labels[i])] = 0
    return out


def crop_rois(rgb, initial_masks, depth, crop_size=CROP_SIZE):
    """One padded ROI per label of initial_masks[0], resized to crop_size (bilinear with
    align_corners=True -- F.upsample_bilinear -- for rgb/depth, nearest for the mask).
    Returns (rgb_crops (N,3,S,S), mask_crops (N,S,S), rois (N,4) x0,y0,x1,y1 inclusive, depth_crops)."""
    _, H, W = initial_masks.shape
    dev = rgb.device
    ids = torch.unique(initial_masks[0])
    ids = ids[ids != 0] if ids.numel() and ids[0] == 0 else ids
    n = ids.shape[0]
    rgb_crops = torch.zeros((n, 3, crop_size, crop_size), device=dev)
    mask_crops = torch.zeros((n, crop_size, crop_size), device=dev)
    depth_crops = torch.zeros((n, 3, crop_size, crop_size), device=dev) if depth is not None else None
    rois = torch.zeros((n, 4), device=dev)
    size = (crop_size, crop_size)
    for k, mask_id in enumerate(ids):
        mask = (initial_masks[0] == mask_id).float()
        x0, y0, x1, y1 = (int(v) for v in mask_to_tight_box(mask))
        # torch.round: half to even, as the reference (test_dataset.py:83-84)
        xp = int(torch.round(torch.tensor(float(x1 - x0)) * PADDING_PERCENTAGE).item())
        yp = int(torch.round(torch.tensor(float(y1 - y0)) * PADDING_PERCENTAGE).item())
        x0, x1 = max(x0 - xp, 0), min(x1 + xp, W - 1)
        y0, y1 = max(y0 - yp, 0), min(y1 + yp, H - 1)
        rois[k] = torch.tensor([x0, y0, x1, y1], dtype=torch.float32)
        rgb_crops[k] = F.interpolate(rgb[0:1, :, y0:y1 + 1, x0:x1 + 1], size=size, mode="bilinear", align_corners=True)[0]
        mask_crops[k] = F.interpolate(mask[None, None, y0:y1 + 1, x0:x1 + 1], size=size, mode="nearest")[0, 0]
        if depth is not None:
            depth_crops[k] = F.interpolate(depth[0:1, :, y0:y1 + 1, x0:x1 + 1], size=size, mode="bilinear",
                                           align_corners=True)[0]
    return rgb_crops, mask_crops, rois, depth_crops


def match_label_crop(initial_masks, labels_crop, out_label_crop, rois, depth_crop):
    """Reject second-stage segments that overlap the first-stage mask by < 50 %, order the crops
    (far-to-near by mean depth, or large-to-small ROI without depth) and paste the renumbered
    segments back at ROI resolution; later crops overwrite earlier ones.
    Returns (refined (1,H,W) float, labels_crop with rejected segments set to -1)."""
    num = labels_crop.shape[0]
    # overlap of every (crop, segment) with the first-stage mask as two histograms (TD:125-131: same counts, same fp32
    # division as the reference's per-segment loop)
    k = int(LABEL_BINS)
    lab = labels_crop.reshape(num, -1).to(torch.int64).clamp(0, k - 1) + torch.arange(num, device=labels_crop.device)[:, None] * k
    area = torch.bincount(lab.reshape(-1), minlength=num * k)[:num * k]
    hit = torch.bincount(lab.reshape(-1), weights=out_label_crop.reshape(-1).float(), minlength=num * k)[:num * k]
    bad = (hit.float() / area.float().clamp_min(1.0) < 0.5) & (area > 0)
    labels_crop[bad[lab].view_as(labels_crop)] = -1
    keys = []
    for i in range(num):
        if depth_crop is not None:
            sel = labels_crop[i] > -1
            z = depth_crop[i, 2][sel] if sel.sum() > 0 else depth_crop[i, 2]
            keys.append((i, torch.mean(z[z > 0])))
        else:
            keys.append((i, (rois[i, 3] - rois[i, 1] + 1) * (rois[i, 2] - rois[i, 0] + 1)))
    order = [i for i, _ in sorted(keys, key=lambda t: t[1], reverse=True)]
    refined = torch.zeros_like(initial_masks).float()
    count = 0
    for i in order:
        ids = torch.unique(labels_crop[i])
        ids = ids[1:] if ids[0] == -1 else ids
        renum = torch.zeros_like(labels_crop[i])
        for mask_id in ids:
            count += 1
            renum[labels_crop[i] == mask_id] = count
        x0, y0, x1, y1 = (int(v) for v in rois[i])
        small = F.interpolate(renum[None, None].float(), size=(y1 - y0 + 1, x1 - x0 + 1), mode="nearest")[0, 0]
        window = refined[0, y0:y1 + 1, x0:x1 + 1]
        nz = small != 0
        window[nz] = small[nz]
    return refined, labels_crop


def nms(masks, scores, thresh):
    """Mask-IoU NMS, kept indices sorted by mask area (lib/fcn/nms.py:3-23).  numpy in / out."""
    flat = masks.reshape(masks.shape[0], -1).astype(np.float32)
    inters = flat @ flat.T
    areas = np.diag(inters)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        inter = inters[i, order[1:]]
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        order = order[np.where(ovr <= thresh)[0] + 1]
    return np.array(keep)[np.argsort(areas[keep]).astype(np.int32)]


def combine_masks_with_NMS(instances):
    """Label image (labels from 2), per-pixel int(score*100) image and (N,5) boxes [x1,y1,x2,y2,score]
    after NMS at 0.7 (lib/fcn/test_utils.py:55-91)."""
    mask = instances.get("pred_masks").to("cpu").numpy()
    scores = instances.get("scores").to("cpu").numpy()
    h, w = (mask.shape[1], mask.shape[2]) if mask.ndim == 3 and mask.shape[0] else instances.image_size
    bin_mask, score_mask = np.zeros((h, w)), np.zeros((h, w))
    if len(mask) == 0:
        return bin_mask, score_mask, np.zeros((0, 5), dtype=np.float32)
    keep = nms(mask, scores, thresh=0.7).astype(int)
    mask, scores = mask[keep], scores[keep]
    bbox = np.zeros((len(mask), 5), dtype=np.float32)
    for k, m in enumerate(mask):
        pos = np.nonzero(m)
        bin_mask[pos] = k + 2
        score_mask[pos] = int(scores[k] * 100)
        bbox[k] = [pos[1].min(), pos[0].min(), pos[1].max(), pos[0].max(), scores[k]]
    return bin_mask, score_mask, bbox


def _labels_from_outputs(outputs, topk, confident_score, low_threshold, num_class, use_nms):
    conf = get_confident_instances(outputs, topk=topk, score=confident_score, num_class=num_class,
                                   low_threshold=low_threshold)
    if use_nms:
        return combine_masks_with_NMS(conf)
    return combine_masks_tensor(conf), None, None     # same values as combine_masks, no host round trip


def test_sample_crop_nolabel(sample, predictor, predictor_crop=None, *, use_depth=True, topk=False,
                             confident_score=0.7, low_threshold=0.4, num_class=2, use_nms=False,
                             depth_threshold=0.5, crop_batch_builder=None):
    """First-stage prediction -> label image -> depth filter -> ROI crops -> second-stage prediction on
    every crop -> paste back (lib/fcn/test_utils.py:339-421).

    sample: {"image_color" (3,H,W), "depth" (3,H,W) xyz (when use_depth), ...}.  `predictor(sample)`
    returns {"instances": Instances}; `predictor_crop` is called ONCE with a list of crop samples
    (batched) when it exposes ``batch_call``, else once per crop."""
    image = sample["image_color"]
    if image.dim() == 4:
        image = image[0]
    sample = dict(sample, image=image, height=image.shape[-2], width=image.shape[-1])
    depth = None
    if use_depth:
        depth = sample["depth"]
        depth = depth[0] if depth.dim() == 4 else depth
    else:
        sample["depth"] = None
    label, score_mask, bbox = _labels_from_outputs(predictor(sample), topk, confident_score, low_threshold, num_class, use_nms)
    dev = image.device
    out_label = torch.as_tensor(label).unsqueeze(0).to(dev)
    out_score = torch.as_tensor(score_mask).unsqueeze(0).to(dev) if score_mask is not None else None
    image4 = image.unsqueeze(0)
    depth4 = depth.unsqueeze(0) if depth is not None else None
    if depth4 is not None:
        thr = 0.8 if "OSD" in str(sample.get("file_name", "")) else depth_threshold      # test_utils.py:384-387
        out_label = filter_labels_depth(out_label, depth4, thr)
    refined = None
    if predictor_crop is not None:
        rgb_crop, out_label_crop, rois, depth_crop = crop_rois(image4, out_label.clone(), depth4)
        n = rgb_crop.shape[0]
        if n > 0:
            crops = [{"image": rgb_crop[i], "height": CROP_SIZE, "width": CROP_SIZE,
                      "depth": depth_crop[i] if depth_crop is not None else None} for i in range(n)]
            outs = predictor_crop.batch_call(crops) if hasattr(predictor_crop, "batch_call") else [predictor_crop(c) for c in crops]
            labels_crop = torch.zeros((n, CROP_SIZE, CROP_SIZE), device=dev)
            for i, o in enumerate(outs):
                lab, _, _ = _labels_from_outputs(o, topk, confident_score, low_threshold, num_class, use_nms)
                labels_crop[i] = torch.as_tensor(lab).to(dev)
            refined, _ = match_label_crop(out_label, labels_crop, out_label_crop, rois, depth_crop)
    return out_label, refined, out_score, bbox
