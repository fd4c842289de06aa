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


LABEL_BINS = 1024      # label images hold 0 and 2..N+1 with N <= detections per image (<= the number of queries)


def mask_to_tight_box(mask):
    """lib/utils/mask.py:179-186: (x_min, y_min, x_max, y_max) of the non-zero pixels."""
    ys, xs = torch.nonzero(mask, as_tuple=True)
    return xs.min(), ys.min(), xs.max(), ys.max()


def label_stats(labels, weight=None):
    """Per-label statistics of integer-valued label images labels (B,H,W) with values in [0, LABEL_BINS):
    (stats (B,k,5) = area, x_min, y_min, x_max, y_max [W, H, -1, -1 when absent]; wsum (B,k) = sum of `weight` over the
    label's pixels; overflow (B,) = number of out-of-range pixels).  One pass replaces the reference's per-label
    unique()/masked reductions (test_dataset.py:62-131, 183-198).  GPU tensors go through the HIP kernel
    (msm_label_stats); CPU tensors -- the host-logic unit tests -- through the same definition in torch ops."""
    k = int(LABEL_BINS)
    B, H, W = labels.shape
    if labels.is_cuda:
        from . import ops
        return ops.label_stats(labels.float().contiguous(), None if weight is None else weight.float().contiguous(), k)
    lab = labels.reshape(B, -1).float()
    idx = lab.to(torch.int64).clamp(0, k - 1)
    overflow = (~(lab >= 0) | (lab.to(torch.int64) >= k)).sum(1).to(torch.int32)
    ys = torch.arange(H).repeat_interleave(W).expand(B, -1)
    xs = torch.arange(W).repeat(H).expand(B, -1)
    area = torch.zeros((B, k), dtype=torch.int64).scatter_add(1, idx, torch.ones_like(idx))
    stats = torch.stack([area,
                         torch.full((B, k), W).scatter_reduce(1, idx, xs, "amin"),
                         torch.full((B, k), H).scatter_reduce(1, idx, ys, "amin"),
                         torch.full((B, k), -1).scatter_reduce(1, idx, xs, "amax"),
                         torch.full((B, k), -1).scatter_reduce(1, idx, ys, "amax")], 2).to(torch.int32)
    wsum = torch.zeros((B, k), dtype=torch.float32)
    if weight is not None:
        wsum.scatter_add_(1, idx, weight.reshape(B, -1).float())
    return stats, wsum, overflow


def filter_labels_depth(labels, depth, threshold):
    """Zero every label whose pixels have valid depth (z > 0) on less than `threshold` of their area.
    labels (B,H,W) with small non-negative integer values, depth (B,3,H,W) xyz.  (lib/fcn/test_dataset.py:183-198; the
    per-label loop of the reference is one statistics pass here: same integer counts, same fp32 division, no host
    syncs.)"""
    stats, good, _ = label_stats(labels, (depth[:, 2] > 0).float())
    area = stats[:, :, 0]
    bad = (good / area.float().clamp_min(1.0) < threshold) & (area > 0)
    bad[:, 0] = False
    idx = labels.reshape(labels.shape[0], -1).to(torch.int64).clamp(0, int(LABEL_BINS) - 1)
    return labels.masked_fill(torch.gather(bad, 1, idx).view_as(labels), 0)


def _label_boxes(label_img):
    """Labels present in an (H,W) label image (0 = background) and their tight boxes, in ascending label order:
    [(label, x_min, y_min, x_max, y_max), ...] -- one statistics pass and ONE device -> host transfer instead of a
    nonzero() + four .item() round trips per label (lib/utils/mask.py:179-186)."""
    stats, _, overflow = label_stats(label_img[None])
    t = torch.cat([stats[0].reshape(-1), overflow]).cpu().numpy()
    if t[-1] != 0:
        raise ValueError(f"label image values must be integers in [0, {LABEL_BINS}) ({t[-1]} pixels are not)")
    t = t[:-1].reshape(-1, 5)
    return [(int(v), int(t[v, 1]), int(t[v, 2]), int(t[v, 3]), int(t[v, 4])) for v in np.nonzero(t[:, 0])[0] if v != 0]


def crop_rois(rgb, initial_masks, depth, crop_size=CROP_SIZE):
    """One padded ROI per label of initial_masks[0], resized to crop_size (bilinear with
    align_corners=True -- F.upsample_bilinear -- for rgb/depth, nearest for the mask).
    Returns (rgb_crops (N,3,S,S), mask_crops (N,S,S), rois (N,4) x0,y0,x1,y1 inclusive, depth_crops)."""
    _, H, W = initial_masks.shape
    dev = rgb.device
    boxes = _label_boxes(initial_masks[0])
    n = len(boxes)
    size = (crop_size, crop_size)
    rois_host, table = [], []
    for mask_id, x0, y0, x1, y1 in boxes:
        # round(): half to even, as the reference's torch.round on the exact product (test_dataset.py:83-84)
        xp, yp = int(round((x1 - x0) * PADDING_PERCENTAGE)), int(round((y1 - y0) * PADDING_PERCENTAGE))
        x0, x1 = max(x0 - xp, 0), min(x1 + xp, W - 1)
        y0, y1 = max(y0 - yp, 0), min(y1 + yp, H - 1)
        rois_host.append([x0, y0, x1, y1])
        table.append([0, int(mask_id), x0, y0, x1, y1, 0, 0])
    rois = torch.tensor(rois_host, dtype=torch.float32).reshape(n, 4).to(dev)
    if rgb.is_cuda and n > 0:
        # device tensors: every crop of the frame in ONE launch of the kernel the batched pipeline uses (msm_crop_resize: bilinear
        # align_corners=True for rgb / depth, nearest for the mask, ATen's index arithmetic -- tests pin it to the loop below)
        from . import ops
        tab = torch.tensor(table, dtype=torch.int32, device=dev)
        rgb_crops, mask_crops, depth_crops = ops.crop_resize(rgb[0:1].float().contiguous(), None if depth is None else depth[0:1].float().contiguous(),
                                                             initial_masks[0:1].float().contiguous(), tab, crop_size)
        return rgb_crops, mask_crops, rois, depth_crops
    # host tensors (unit tests of the harness logic without a GPU): the reference's per-ROI loop
    rgb_crops = torch.zeros((n, 3, crop_size, crop_size), device=dev)
    mask_crops = torch.zeros((n, crop_size, crop_size), device=dev)
    depth_crops = torch.zeros((n, 3, crop_size, crop_size), device=dev) if depth is not None else None
    for k, (mask_id, x0, y0, x1, y1) in enumerate([(t[1], *t[2:6]) for t in table]):
        mask = (initial_masks[0, y0:y1 + 1, x0:x1 + 1] == mask_id).float()
        rgb_crops[k] = F.interpolate(rgb[0:1, :, y0:y1 + 1, x0:x1 + 1], size=size, mode="bilinear", align_corners=True)[0]
        mask_crops[k] = F.interpolate(mask[None, None], size=size, mode="nearest")[0, 0]
        if depth is not None:
            depth_crops[k] = F.interpolate(depth[0:1, :, y0:y1 + 1, x0:x1 + 1], size=size, mode="bilinear",
                                           align_corners=True)[0]
    return rgb_crops, mask_crops, rois, depth_crops


def match_label_crop(initial_masks, labels_crop, out_label_crop, rois, depth_crop):
    """Reject second-stage segments that overlap the first-stage mask by < 50 %, order the crops
    (far-to-near by mean depth, or large-to-small ROI without depth) and paste the renumbered
    segments back at ROI resolution; later crops overwrite earlier ones.
    Returns (refined (1,H,W) float, labels_crop with rejected segments set to -1).

    The reference's per-crop / per-segment loops (test_dataset.py:116-179) are table lookups here: two histograms for
    the overlap test, one (crop, label) -> new number table for the renumbering, and two host transfers in all (the
    sort keys and the ROIs).  The mean depth of a crop is accumulated in fp64 (the reference's fp32 torch.mean can
    order two crops whose mean depths agree to ~1e-7 either way)."""
    num = labels_crop.shape[0]
    dev = labels_crop.device
    refined = torch.zeros_like(initial_masks).float()
    if num == 0:
        return refined, labels_crop
    k = int(LABEL_BINS)
    stats, hit, _ = label_stats(labels_crop, out_label_crop)
    area = stats[:, :, 0].reshape(-1)
    lab = labels_crop.reshape(num, -1).to(torch.int64).clamp(0, k - 1) + torch.arange(num, device=dev)[:, None] * k
    bad = (hit.reshape(-1) / area.float().clamp_min(1.0) < 0.5) & (area > 0)
    labels_crop.masked_fill_(bad[lab].view_as(labels_crop), -1)
    rois_host = [[int(v) for v in r] for r in rois.tolist()]
    if depth_crop is not None:
        sel = (labels_crop > -1).reshape(num, -1)
        z = depth_crop[:, 2].reshape(num, -1)
        use = (sel | ~sel.any(1, keepdim=True)) & (z > 0)
        keys = ((z * use).sum(1, dtype=torch.float64) / use.sum(1)).tolist()            # 0/0 = nan like mean of nothing
    else:
        keys = [float((r[3] - r[1] + 1) * (r[2] - r[0] + 1)) for r in rois_host]
    order = [i for i, _ in sorted(enumerate(keys), key=lambda t: t[1], reverse=True)]
    # new numbers 1.. in (crop order, ascending surviving label) order
    order_t = torch.tensor(order, device=dev)
    alive = ((area > 0) & ~bad).view(num, k)[order_t]
    number = torch.zeros((num, k), dtype=torch.float32, device=dev)
    number[order_t] = (torch.cumsum(alive.reshape(-1), 0).view(num, k) * alive).float()
    renum = number.view(-1)[lab].view(num, 1, *labels_crop.shape[1:])
    for i in order:
        x0, y0, x1, y1 = rois_host[i]
        small = F.interpolate(renum[i:i + 1], size=(y1 - y0 + 1, x1 - x0 + 1), mode="nearest")[0, 0]
        window = refined[0, y0:y1 + 1, x0:x1 + 1]
        window.copy_(torch.where(small != 0, small, window))
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


def label_image(outputs, topk, confident_score, low_threshold, num_class):
    """combine_masks(get_confident_instances(outputs, ...)) (test_utils.py:35-53, 93-112) without selecting the
    instances first: instance i, if kept, carries label 2 + (number of kept instances before it), and "later
    instances overwrite earlier ones" is the per-pixel maximum of those labels.  Same values as the two reference
    steps, no data-dependent shapes, so no device -> host round trip.  Returns an (H,W) float64 tensor."""
    inst = outputs["instances"]
    masks, scores = inst.get("pred_masks"), inst.get("scores")
    if masks.dim() != 3 or masks.shape[0] == 0:
        h, w = inst.image_size
        return torch.zeros((h, w), dtype=torch.float64, device=scores.device)
    if topk:
        keep = ((inst.get("pred_classes") == 1) & (scores > low_threshold)) if num_class >= 2 else torch.ones_like(scores, dtype=torch.bool)
    else:
        keep = scores > confident_score
    lab = ((torch.cumsum(keep, 0) + 1) * keep).to(torch.int16)
    return ((masks != 0).to(torch.int16) * lab[:, None, None]).amax(0).to(torch.float64)


def _labels_from_outputs(outputs, topk, confident_score, low_threshold, num_class, use_nms):
    if use_nms:
        conf = get_confident_instances(outputs, topk=topk, score=confident_score, num_class=num_class,
                                       low_threshold=low_threshold)
        return combine_masks_with_NMS(conf)
    return label_image(outputs, topk, confident_score, low_threshold, num_class), None, None


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


# ----------------------------------------------------------------------------------------------------------------------
# The same pipeline for a BATCH of frames (BASELINE configs[3]: batch = 16): lib/fcn/test_utils.py:375-406 is a serial
# loop over frames and, inside it, over crops (batch 1 each).  Nothing couples two frames, so here the first stage runs
# on all frames in one call, every frame's ROIs are cut in one launch (ops.crop_resize), all crops of all frames go
# through the second stage in batches of `crop_batch` (measured at 171 crops: one call 20.8 ms, three calls of <= 64 22.2 ms), and every frame's refined labels are pasted in one launch
# (ops.paste_labels).  Two device -> host transfers per batch in all (the label statistics that define the ROIs, the
# depth keys that order the paste).  Per frame the results are those of test_sample_crop_nolabel (non-NMS form).
# ----------------------------------------------------------------------------------------------------------------------
def _batch_tensors(predictor, samples):
    """(scores (B,K), classes (B,K), masks (B,K,H,W)) of a list of samples: a predictor that exposes ``batch_tensors`` hands the
    batched tensors of its model over as they are; otherwise the per-sample Instances are stacked (a copy)."""
    if hasattr(predictor, "batch_tensors"):
        return predictor.batch_tensors(samples)
    outs = predictor.batch_call(samples) if hasattr(predictor, "batch_call") else [predictor(s) for s in samples]
    inst = [o["instances"] for o in outs]
    return (torch.stack([i.get("scores") for i in inst]), torch.stack([i.get("pred_classes") for i in inst]),
            torch.stack([i.get("pred_masks") for i in inst]))


def instance_labels(scores, classes, topk, confident_score, low_threshold, num_class):
    """The label each instance carries in the label image (label_image above, batched): 2 + (kept instances before it), 0
    when the instance is dropped (get_confident_instances, test_utils.py:35-52).  scores / classes (B,K) -> (B,K) float."""
    if topk:
        keep = ((classes == 1) & (scores > low_threshold)) if num_class >= 2 else torch.ones_like(scores, dtype=torch.bool)
    else:
        keep = scores > confident_score
    return ((torch.cumsum(keep, 1) + 1) * keep).float()


def roi_table(stats, overflow, H, W):
    """Host side of crop_rois for a batch: stats (F,k,5) / overflow (F,) as numpy (ONE transfer) -> list of rows
    [frame, label, x0, y0, x1, y1, 0, 0] in (frame, ascending label) order, boxes padded by 25 % and clipped
    (test_dataset.py:76-92; round half to even like torch.round)."""
    if overflow.any():
        raise ValueError(f"label image values must be integers in [0, {LABEL_BINS})")
    rows = []
    for f in range(stats.shape[0]):
        t = stats[f]
        for v in np.nonzero(t[:, 0])[0]:
            if v == 0:
                continue
            x0, y0, x1, y1 = (int(t[v, 1]), int(t[v, 2]), int(t[v, 3]), int(t[v, 4]))
            xp, yp = int(round((x1 - x0) * PADDING_PERCENTAGE)), int(round((y1 - y0) * PADDING_PERCENTAGE))
            rows.append([f, int(v), max(x0 - xp, 0), max(y0 - yp, 0), min(x1 + xp, W - 1), min(y1 + yp, H - 1), 0, 0])
    return rows


# GPU tensors go through the HIP kernels; CPU tensors -- the host-logic unit tests -- through the same definitions in torch ops
# (the reference's own per-ROI F.interpolate calls), like label_stats above.
def _label_image_batched(masks, inst_labels):
    if masks.is_cuda:
        from . import ops
        return ops.label_image(masks.float().contiguous(), inst_labels.float().contiguous())
    return ((masks != 0).float() * inst_labels[:, :, None, None].float()).amax(1) if masks.shape[1] else masks.new_zeros((masks.shape[0],) + masks.shape[2:])


def _crop_resize_batched(images, depths, labels, rows, size):
    if images.is_cuda:
        from . import ops
        return ops.crop_resize(images, depths, labels, torch.tensor(rows, dtype=torch.int32, device=images.device), size)
    n = len(rows)
    rgb = torch.zeros((n, 3, size, size))
    msk = torch.zeros((n, size, size))
    dep = torch.zeros((n, 3, size, size)) if depths is not None else None
    for k, (f, lab, x0, y0, x1, y1, _, _) in enumerate(rows):
        rgb[k] = F.interpolate(images[f:f + 1, :, y0:y1 + 1, x0:x1 + 1], size=(size, size), mode="bilinear", align_corners=True)[0]
        msk[k] = F.interpolate((labels[f, y0:y1 + 1, x0:x1 + 1] == lab).float()[None, None], size=(size, size), mode="nearest")[0, 0]
        if depths is not None:
            dep[k] = F.interpolate(depths[f:f + 1, :, y0:y1 + 1, x0:x1 + 1], size=(size, size), mode="bilinear", align_corners=True)[0]
    return rgb, msk, dep


def _paste_batched(renum, rows, order, frame_start, frames, H, W):
    if renum.is_cuda:
        from . import ops
        dev = renum.device
        return ops.paste_labels(renum, torch.tensor(rows, dtype=torch.int32, device=dev), torch.tensor(order, dtype=torch.int32, device=dev),
                                torch.tensor(frame_start, dtype=torch.int32, device=dev), frames, H, W)
    refined = torch.zeros((frames, H, W))
    for f in range(frames):
        for n in order[frame_start[f]:frame_start[f + 1]]:
            _, _, x0, y0, x1, y1, _, _ = rows[n]
            small = F.interpolate(renum[n][None, None], size=(y1 - y0 + 1, x1 - x0 + 1), mode="nearest")[0, 0]
            window = refined[f, y0:y1 + 1, x0:x1 + 1]
            window.copy_(torch.where(small != 0, small, window))
    return refined


def _match_pre(labels_crop, out_label_crop, depth_crop):
    """Device half of match_label_crop_batched, before the paste order is known: the overlap test (second-stage segments that cover
    the first-stage mask by < 50 % are set to -1 in ``labels_crop``, in place) and the crops' sort keys (mean valid depth of the
    surviving segments, fp64; None without depth).  -> (area (N*k,), bad (N*k,), lab (N, S*S) indices into them, keys (N,) or None).
    No host transfer, no data-dependent shape: this half is captured in the second-stage HIP graph of BatchedTwoStage."""
    num = labels_crop.shape[0]
    k = int(LABEL_BINS)
    stats, hit, _ = label_stats(labels_crop, out_label_crop)
    area = stats[:, :, 0].reshape(-1)
    lab = labels_crop.reshape(num, -1).to(torch.int64).clamp(0, k - 1) + torch.arange(num, device=labels_crop.device)[:, None] * k
    bad = (hit.reshape(-1) / area.float().clamp_min(1.0) < 0.5) & (area > 0)
    labels_crop.masked_fill_(bad[lab].view_as(labels_crop), -1)
    keys = None
    if depth_crop is not None:
        sel = (labels_crop > -1).reshape(num, -1)
        z = depth_crop[:, 2].reshape(num, -1)
        use = (sel | ~sel.any(1, keepdim=True)) & (z > 0)
        keys = (z * use).sum(1, dtype=torch.float64) / use.sum(1)                       # 0/0 = nan like mean of nothing
    return area, bad, lab, keys


def _match_post(area, bad, lab, keys, rows, frames, H, W, crop_shape):
    """Host-ordered half: ``keys`` (list of floats, one per crop of ``rows``) -> paste order inside every frame, renumbering, paste.
    area / bad / lab may describe MORE crops than len(rows) (a padded second-stage batch): only the first len(rows) are used."""
    num = len(rows)
    k = int(LABEL_BINS)
    dev = lab.device
    area, bad, lab = area.view(-1, k)[:num].reshape(-1), bad.view(-1, k)[:num].reshape(-1), lab[:num]
    frame_of = [r[0] for r in rows]
    # paste order inside a frame: descending key, ties and NaN exactly as sorted(reverse=True) leaves them in match_label_crop
    order, frame_start = [], [0]
    by_frame = [[] for _ in range(frames)]
    for n, f in enumerate(frame_of):
        by_frame[f].append(n)
    for mine in by_frame:
        order += [mine[i] for i, _ in sorted(enumerate([keys[n] for n in mine]), key=lambda t: t[1], reverse=True)]
        frame_start.append(len(order))
    order_t = torch.tensor(order, device=dev)
    alive = ((area > 0) & ~bad).view(num, k)[order_t]                                    # rows in (frame, paste) order
    c = torch.cumsum(alive.reshape(-1), 0).view(num, k)
    # numbers restart at 1 in every frame: subtract what was counted before the frame's first crop
    before = torch.cat([c.new_zeros(1), c[:, -1]])[torch.tensor([frame_start[frame_of[n]] for n in order], device=dev)]
    number = torch.zeros((num, k), dtype=torch.float32, device=dev)
    number[order_t] = ((c - before[:, None]) * alive).float()
    renum = number.view(-1)[lab].view(num, *crop_shape).contiguous()
    return _paste_batched(renum, rows, order, frame_start, frames, H, W)


def match_label_crop_batched(initial_masks, labels_crop, out_label_crop, rows, depth_crop):
    """match_label_crop for the crops of a whole batch of frames: initial_masks (F,H,W), labels_crop / out_label_crop
    (N,S,S), rows = roi_table(...) (crop n belongs to frame rows[n][0]), depth_crop (N,3,S,S) or None.
    Returns refined (F,H,W).  Same arithmetic per frame as match_label_crop; the renumbering restarts at 1 in every frame."""
    Fr, H, W = initial_masks.shape
    if labels_crop.shape[0] == 0:
        return torch.zeros_like(initial_masks).float()
    area, bad, lab, keys = _match_pre(labels_crop, out_label_crop, depth_crop)
    if keys is not None:
        keys = keys.tolist()                                                             # the batch's second (last) transfer
    else:
        keys = [float((r[5] - r[3] + 1) * (r[4] - r[2] + 1)) for r in rows]
    return _match_post(area, bad, lab, keys, rows, Fr, H, W, tuple(labels_crop.shape[1:]))


def test_batch_crop_nolabel(samples, predictor, predictor_crop=None, *, use_depth=True, topk=False, confident_score=0.7,
                            low_threshold=0.4, num_class=2, depth_threshold=0.5, crop_batch=256, stages=None):
    """test_sample_crop_nolabel (non-NMS form) for a list of frames of one size, batched end to end.
    samples: [{"image_color" (3,H,W), "depth" (3,H,W), ...}, ...] on the GPU.  Returns (out_label (F,H,W), refined (F,H,W) or
    None, rows) -- frame f's results equal test_sample_crop_nolabel(samples[f], ...)[0][0] / [1][0]; ``rows`` is the ROI table
    (frame, label, x0, y0, x1, y1, 0, 0) of the second stage.  ``stages``: a dict that receives the intermediate tensors
    (crops, second-stage label images) -- for tests."""
    images = torch.stack([s["image_color"][0] if s["image_color"].dim() == 4 else s["image_color"] for s in samples]).float().contiguous()
    Fr, _, H, W = images.shape
    depths = None
    if use_depth:
        depths = torch.stack([s["depth"][0] if s["depth"].dim() == 4 else s["depth"] for s in samples]).float().contiguous()
    first = [{"image": images[f], "depth": depths[f] if depths is not None else None, "height": H, "width": W} for f in range(Fr)]
    kw = dict(topk=topk, confident_score=confident_score, low_threshold=low_threshold, num_class=num_class)
    scores, classes, masks = _batch_tensors(predictor, first)
    out_label = _label_image_batched(masks, instance_labels(scores, classes, **kw))
    if depths is not None:
        thr = torch.tensor([0.8 if "OSD" in str(s.get("file_name", "")) else depth_threshold for s in samples],
                           device=images.device, dtype=torch.float32)[:, None]          # test_utils.py:384-387
        out_label = filter_labels_depth(out_label, depths, thr)
    if predictor_crop is None:
        return out_label, None, []
    stats, _, overflow = label_stats(out_label)
    packed = torch.cat([stats.reshape(-1), overflow]).cpu().numpy()                      # the batch's first transfer
    rows = roi_table(packed[:-Fr].reshape(Fr, -1, 5), packed[-Fr:], H, W)
    n = len(rows)
    if n == 0:
        return out_label, torch.zeros_like(out_label), rows
    rgb_crop, mask_crop, depth_crop = _crop_resize_batched(images, depths, out_label, rows, CROP_SIZE)
    labels_crop = torch.empty((n, CROP_SIZE, CROP_SIZE), device=images.device, dtype=torch.float32)
    for c0 in range(0, n, crop_batch):
        c1 = min(n, c0 + crop_batch)
        crops = [{"image": rgb_crop[i], "height": CROP_SIZE, "width": CROP_SIZE,
                  "depth": depth_crop[i] if depth_crop is not None else None} for i in range(c0, c1)]
        s2, k2, m2 = _batch_tensors(predictor_crop, crops)
        labels_crop[c0:c1] = _label_image_batched(m2, instance_labels(s2, k2, **kw))
    if stages is not None:
        stages.update(rgb_crop=rgb_crop, mask_crop=mask_crop, depth_crop=depth_crop, labels_crop=labels_crop.clone())
    refined = match_label_crop_batched(out_label, labels_crop, mask_crop, rows, depth_crop)
    return out_label, refined, rows


# ----------------------------------------------------------------------------------------------------------------------
# configs[3] as a replayable pipeline: both stages from captured HIP graphs, the two device -> host transfers of a batch
# overlapped with the other batch in flight.
# ----------------------------------------------------------------------------------------------------------------------
class BatchedTwoStage:
    """test_batch_crop_nolabel (non-NMS form, lib/fcn/test_utils.py:339-421 per frame) for batches of ``frames`` frames of one
    size, with the model called directly (``model.inference_images``: backbone + head + post-processing) and both stages replayed
    from HIP graphs:

      graph 1   frames -> model -> label images -> depth filter -> label statistics -> pinned host buffer   (fixed shapes)
      host      ROI table from the statistics (test_dataset.py:76-92), uploaded into the slot's table buffer
      graph 2   (one per crop-count BUCKET: the N crops of a batch are padded to the next multiple of ``bucket`` (16) with copies of
                crop 0, whose results are ignored) every ROI cut and resized -> model -> crop label images -> overlap test ->
                the crops' depth keys -> pinned host buffer
      host      paste order from the keys; renumbering + paste-back launches (eager: their shapes depend on N)

    ``run(batches)`` keeps TWO batches in flight (two slots, one stream each): while the host waits for one slot's statistics
    or keys, the GPU works on the other slot.  ``__call__(samples)`` runs one batch on slot 0.  Per frame the results are those
    of test_batch_crop_nolabel up to the batch-size dependence of the head's reduction orders (a padded second-stage batch is a
    different batch size: tests hold the pipeline to the same oracle bounds as the eager form).

    The captured graphs follow the model's execution plan: a plan switch (set_precision, ...) or a parameter update re-captures
    (graphs.StaleCheck)."""

    def __init__(self, model, frames, size, *, use_depth=True, topk=False, confident_score=0.7, low_threshold=0.4, num_class=2,
                 depth_threshold=0.5, bucket=16, crop_size=CROP_SIZE, slots=2, graphs=True):
        from .graphs import StaleCheck, _slot_stream
        self.model = model
        self.frames, (self.H, self.W) = int(frames), (int(size[0]), int(size[1]))
        self.use_depth = bool(use_depth)
        self.kw = dict(topk=topk, confident_score=confident_score, low_threshold=low_threshold, num_class=num_class)
        self.depth_threshold = float(depth_threshold)
        self.bucket, self.crop_size, self.use_graphs = int(bucket), int(crop_size), bool(graphs)
        self._sig = StaleCheck(model)
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("BatchedTwoStage needs the model on the GPU (there is no CPU path)")
        self.dev = dev
        self._slots = [self._new_slot(_slot_stream(dev, i)) for i in range(max(1, int(slots)))]

    # ---- slot state ----
    def _new_slot(self, stream):
        Fr, H, W, dev = self.frames, self.H, self.W, self.dev
        k = int(LABEL_BINS)
        with torch.cuda.stream(stream):
            st = dict(stream=stream, sig=None, g1=None, g2={}, label=None, s2={},
                      images=torch.zeros((Fr, 3, H, W), device=dev), depths=torch.zeros((Fr, 3, H, W), device=dev) if self.use_depth else None,
                      thr=torch.full((Fr, 1), self.depth_threshold, device=dev), thr_host=[self.depth_threshold] * Fr,
                      host_stats=torch.zeros(Fr * k * 5 + Fr, dtype=torch.int32).pin_memory(),
                      ev1=torch.cuda.Event(), ev2=torch.cuda.Event(), rows=None, n=0, nb=0)
        return st

    def _predict(self, images, depths):
        inputs = {"image": images}
        if depths is not None:
            inputs["depth"] = depths
        sc, cl, mk = self.model.inference_images(inputs, tuple(int(v) for v in images.shape[-2:]))[:3]
        return sc, cl, mk

    def _stage1(self, st):
        sc, cl, mk = self._predict(st["images"], st["depths"])
        lab = _label_image_batched(mk, instance_labels(sc, cl, **self.kw))
        if st["depths"] is not None:
            lab = filter_labels_depth(lab, st["depths"], st["thr"])
        stats, _, overflow = label_stats(lab)
        st["host_stats"].copy_(torch.cat([stats.reshape(-1), overflow]), non_blocking=True)          # the batch's first transfer
        return lab

    def _stage2(self, st, nb):
        from . import ops
        b = st["s2"][nb]
        rgb, msk, dep = ops.crop_resize(st["images"], st["depths"], st["label"], b["table"], self.crop_size)
        sc, cl, mk = self._predict(rgb, dep)
        labels_crop = _label_image_batched(mk, instance_labels(sc, cl, **self.kw))
        raw = labels_crop.clone()
        area, bad, lab, keys = _match_pre(labels_crop, msk, dep)
        if keys is not None:
            b["host_keys"].copy_(keys, non_blocking=True)                                             # the batch's second transfer
        return dict(rgb_crop=rgb, mask_crop=msk, depth_crop=dep, labels_crop=raw, area=area, bad=bad, lab=lab)

    def _capture(self, st, fn, *args):
        """Run fn twice (weight caches, MIOpen's solver choices), then capture it on the slot's stream."""
        if not self.use_graphs:
            return None, fn(st, *args)
        for _ in range(2):
            fn(st, *args)
        st["stream"].synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st["stream"]):
            out = fn(st, *args)
        from .graphs import cache_refs
        st["refs"].append(cache_refs(self.model))              # the derived tensors the graph reads by address stay alive with it
        return g, out

    # ---- the phases of one batch on one slot (all device work on the slot's stream) ----
    @torch.no_grad()
    def _phase1(self, st, samples):
        if len(samples) != self.frames:
            raise ValueError(f"BatchedTwoStage was built for {self.frames} frames, got {len(samples)}")
        sig = self._sig()
        if st["sig"] != sig:                                   # first use, plan switch or parameter update: re-capture everything
            st["stream"].synchronize()
            st.update(sig=sig, g1=None, g2={}, s2={}, label=None, refs=[])
        with torch.cuda.stream(st["stream"]):
            st["stream"].wait_stream(torch.cuda.current_stream())
            torch.stack([s["image_color"][0] if s["image_color"].dim() == 4 else s["image_color"] for s in samples], out=st["images"])
            if st["depths"] is not None:
                torch.stack([s["depth"][0] if s["depth"].dim() == 4 else s["depth"] for s in samples], out=st["depths"])
                thr = [0.8 if "OSD" in str(s.get("file_name", "")) else self.depth_threshold for s in samples]      # test_utils.py:384-387
                if thr != st["thr_host"]:
                    st["thr"].copy_(torch.tensor(thr, dtype=torch.float32)[:, None])
                    st["thr_host"] = thr
            if st["label"] is None:
                st["g1"], st["label"] = self._capture(st, self._stage1)
                if st["g1"] is not None:
                    st["g1"].replay()
            elif st["g1"] is not None:
                st["g1"].replay()
            else:
                st["label"] = self._stage1(st)
            st["ev1"].record(st["stream"])

    @torch.no_grad()
    def _phase2(self, st):
        Fr, k = self.frames, int(LABEL_BINS)
        st["ev1"].synchronize()
        packed = st["host_stats"].numpy()
        rows = roi_table(packed[:Fr * k * 5].reshape(Fr, k, 5), packed[Fr * k * 5:], self.H, self.W)
        st["rows"], st["n"] = rows, len(rows)
        if not rows:
            return
        nb = -(-len(rows) // self.bucket) * self.bucket
        st["nb"] = nb
        with torch.cuda.stream(st["stream"]):
            if nb not in st["s2"]:
                st["s2"][nb] = dict(table=torch.zeros((nb, 8), dtype=torch.int32, device=self.dev),
                                    table_host=torch.zeros((nb, 8), dtype=torch.int32).pin_memory(),
                                    host_keys=torch.zeros(nb, dtype=torch.float64).pin_memory(), out=None)
            b = st["s2"][nb]
            b["table_host"].copy_(torch.tensor(rows + [rows[0]] * (nb - len(rows)), dtype=torch.int32))
            b["table"].copy_(b["table_host"], non_blocking=True)
            if b["out"] is None:
                st["g2"][nb], b["out"] = self._capture(st, self._stage2, nb)
                if st["g2"][nb] is not None:
                    st["g2"][nb].replay()
            elif st["g2"][nb] is not None:
                st["g2"][nb].replay()
            else:
                b["out"] = self._stage2(st, nb)
            st["ev2"].record(st["stream"])

    @torch.no_grad()
    def _phase3(self, st, stages=None):
        """-> (out_label (F,H,W), refined (F,H,W), rows): tensors owned by the slot (valid until its next batch)."""
        label, rows, n = st["label"], st["rows"], st["n"]
        if n == 0:
            return label, torch.zeros_like(label), rows
        st["ev2"].synchronize()
        b = st["s2"][st["nb"]]
        o = b["out"]
        if self.use_depth:
            keys = b["host_keys"].numpy()[:n].tolist()
        else:
            keys = [float((r[5] - r[3] + 1) * (r[4] - r[2] + 1)) for r in rows]
        with torch.cuda.stream(st["stream"]):
            refined = _match_post(o["area"], o["bad"], o["lab"], keys, rows, self.frames, self.H, self.W, (self.crop_size, self.crop_size))
        torch.cuda.current_stream().wait_stream(st["stream"])
        if stages is not None:
            stages.update(rgb_crop=o["rgb_crop"][:n], mask_crop=o["mask_crop"][:n], depth_crop=None if o["depth_crop"] is None else o["depth_crop"][:n],
                          labels_crop=o["labels_crop"][:n].clone())
        return label, refined, rows

    def __call__(self, samples, stages=None):
        st = self._slots[0]
        self._phase1(st, samples)
        self._phase2(st)
        return self._phase3(st, stages)

    def run(self, batches, consume=None):
        """Every batch of ``batches`` (lists of ``frames`` samples) through the pipeline with two batches in flight.  ``consume(i,
        out_label, refined, rows)`` is called per batch while the slot still owns the tensors; without it the results are cloned
        into the returned list."""
        S = len(self._slots)
        results = [None] * len(batches)

        def finish(i):
            out = self._phase3(self._slots[i % S])
            if consume is not None:
                consume(i, *out)
            else:
                results[i] = (out[0].clone(), out[1].clone(), out[2])

        for i, samples in enumerate(batches):
            if i >= 1:
                self._phase2(self._slots[(i - 1) % S])          # batch i-1's statistics -> its second stage is queued ...
            if i >= S:
                finish(i - S)                                   # ... before the host waits for batch i-S's keys (whose slot batch i reuses)
            self._phase1(self._slots[i % S], samples)
        if batches:
            self._phase2(self._slots[(len(batches) - 1) % S])
        for i in range(max(0, len(batches) - S), len(batches)):
            finish(i)
        return None if consume is not None else results
