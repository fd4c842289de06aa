"""Classic UCN vMF mean-shift clustering on the HIP kernels.

Mirrors lib/utils/mean_shift.py (and its copy under MSMFormer/.../transformer_decoder/mean_shift.py)
and the driver lib/fcn/test_dataset.py:44-59: same function names, argument meaning and return
values.  The O(n) passes (seeding, hill climbing, assignment, relabel) run on the GPU; the
order-dependent merge of <= a few hundred seeds (connected_components, mean_shift.py:41-76) runs
on the host exactly as the reference does, on a (S, 64) copy of the seeds.

Only the cosine metric is implemented (cfg.TRAIN.EMBEDDING_METRIC == 'cosine' in every shipped
experiment config); 'euclidean' raises.
"""
import numpy as np
import torch

from . import ops

EMBEDDING_ALPHA = 0.02   # cfg.TRAIN.EMBEDDING_ALPHA, lib/fcn/config.py:255


def _cosine_only(metric):
    if metric != "cosine":
        raise NotImplementedError("only metric='cosine' is implemented on the HIP path")


def ball_kernel(Z, X, kappa, metric="cosine"):
    """mean_shift.py:11-27.  Materialises the (S, n) kernel matrix -- provided for API parity and
    small problems only; the clustering path below never builds it."""
    _cosine_only(metric)
    return torch.exp(kappa * ops.gemm(Z.contiguous(), X.contiguous()))


def get_label_mode(array):
    labels, counts = np.unique(array, return_counts=True)
    return labels[np.argmax(counts)].item()


def connected_components(Z, epsilon, metric="cosine"):
    """mean_shift.py:41-76: sequential, order-dependent merge of the converged seeds.  On the device (one wave,
    ops.ms_connected_components) for up to 304 seeds -- no host synchronisation; larger seed sets take the host loop."""
    _cosine_only(metric)
    if Z.is_cuda and Z.shape[0] <= 304 and Z.shape[1] == 64:
        return ops.ms_connected_components(Z.contiguous(), epsilon)[0]
    return connected_components_host(Z, epsilon)


def connected_components_host(Z, epsilon):
    """The same merge on the host, on a copy of the seeds (the reference's own form; fallback and test partner)."""
    Zc = Z.detach().to("cpu", torch.float32)
    n = Zc.shape[0]
    K = 0
    labels = np.full((n,), -1, dtype=np.int64)          # bookkeeping in numpy: ~100 seeds, a dozen components
    for i in range(n):
        if labels[i] != -1:
            continue
        comp = ((0.5 * (1 - Zc @ Zc[i])) <= epsilon).numpy()     # same fp32 matrix-vector product as the reference
        seen = labels[comp]
        if np.unique(seen).shape[0] > 1:
            label = get_label_mode(seen[seen != -1])
        else:
            label = K
            K += 1
        labels[comp] = label
    return torch.from_numpy(labels)


def seed_hill_climbing_ball(X, Z, kappa, max_iters=10, metric="cosine", precision="f32", xb=None):
    """mean_shift.py:79-109.  ``precision`` (not in the reference): "f32" (fp32 MFMAs), "f32_split" (fp32 results on
    the bf16 matrix pipe) or "bf16" (single bf16 products over a bf16 copy of X, BASELINE configs[4]); ops.ms_hill_climb."""
    _cosine_only(metric)
    return ops.ms_hill_climb(X.contiguous(), Z.contiguous(), kappa, max_iters, precision=precision, xb=xb)


def mean_shift_with_seeds(X, Z, kappa, max_iters=10, metric="cosine", precision="f32"):
    Z = seed_hill_climbing_ball(X, Z, kappa, max_iters=max_iters, metric=metric, precision=precision)
    return connected_components(Z, 2 * EMBEDDING_ALPHA, metric=metric), Z


def _components_with_count(Z, epsilon):
    """(seed_labels, num) with num = len(unique(seed_labels)) as a 1-element int32 device tensor (no host synchronisation on
    the device path): what mean_shift.py:211-216 bounds its per-label counts by."""
    if Z.is_cuda and Z.shape[0] <= 304 and Z.shape[1] == 64:
        labels, num = ops.ms_connected_components(Z.contiguous(), epsilon)
        return labels, num[:1]
    labels = connected_components_host(Z, epsilon)
    return labels, torch.tensor([int(torch.unique(labels).numel())], dtype=torch.int32, device=Z.device)


def select_smart_seeds(X, num_seeds, return_selected_indices=False, init_seeds=None, num_init_seeds=None,
                       metric="cosine", first_index=None, stepwise=False, xb=None):
    """mean_shift.py:128-189.  The first seed index comes from np.random.randint(0, n) like the
    reference (mean_shift.py:155) unless ``first_index`` is given."""
    _cosine_only(metric)
    if init_seeds is not None:
        raise NotImplementedError("init_seeds is unused by the inference path")
    if first_index is None:
        first_index = np.random.randint(0, X.shape[0])
    seeds, idx = ops.ms_select_seeds(X.contiguous(), num_seeds, int(first_index), stepwise=stepwise, xb=xb)
    if not return_selected_indices:
        # the caller cannot see the indices, so the give-up of the persistent kernel (ops.ms_select_seeds) is handled here
        if not stepwise and int(idx.min()) < 0:
            seeds, idx = ops.ms_select_seeds(X.contiguous(), num_seeds, int(first_index), stepwise=True, xb=xb)
        return (seeds,)
    return seeds, idx


def mean_shift_smart_init(X, kappa, num_seeds=100, max_iters=10, metric="cosine", first_index=None, precision="f32"):
    """mean_shift.py:192-229.  Returns (cluster_labels (n,) int64 on X.device, selected_indices (S,)).  ``precision`` "f32" /
    "f32_split": the reference's arithmetic (exact labels); "bf16" (BASELINE configs[4]): seeding and hill climb stream one bf16 copy
    of X -- the same clusters, possibly other member points as seeds and another numbering of the labels."""
    X = X.contiguous()
    if first_index is None:
        first_index = np.random.randint(0, X.shape[0])                   # MS:155, drawn once: a retry reuses it
    xb = ops.ms_pack_bf16(X) if precision == "bf16" else None
    seeds, selected = select_smart_seeds(X, num_seeds, return_selected_indices=True, metric=metric,
                                         first_index=first_index, xb=xb)
    def rest(seeds):
        _cosine_only(metric)
        Z = seed_hill_climbing_ball(X, seeds, kappa, max_iters=max_iters, metric=metric, precision=precision, xb=xb)
        seed_labels, num = _components_with_count(Z, 2 * EMBEDDING_ALPHA)
        # labels are created in order 0, 1, ...: at most one per seed, so num_seeds bounds the histogram of the assignment; the
        # largest-cluster swap looks at labels 0 .. len(unique(seed_labels)) - 1 only, like the reference (MS:211-222)
        labels, counts = ops.ms_assign(X, Z, seed_labels.to(X.device), seeds.shape[0])
        ops.ms_relabel_largest_zero(labels, counts, num)
        return labels

    labels = rest(seeds)
    # The whole clustering is queued without a host synchronisation (the merge of the seeds runs on the device); the one
    # check that needs the host comes last, when everything is in flight: the give-up flag of the persistent seeding kernel
    # (co-residency lost to other streams / processes).  A give-up re-runs seeding on the one-launch-per-step path
    # (identical results) and everything after it.
    if int(selected.min()) < 0:
        seeds, selected = select_smart_seeds(X, num_seeds, return_selected_indices=True, metric=metric,
                                             first_index=first_index, stepwise=True, xb=xb)
        labels = rest(seeds)
    return labels, selected


def clustering_features(features, num_seeds=100, metric="cosine", precision="f32"):
    """lib/fcn/test_dataset.py:44-59: features (B,C,H,W) unit-norm along C -> (out_label (B,H,W) float,
    selected_pixels list of (S,) index tensors).  kappa=20, 10 iterations."""
    B, C, H, W = features.shape
    out_label = torch.zeros((B, H, W), device=features.device)
    selected_pixels = []
    for j in range(B):
        X = ops.transpose_last2(features[j].reshape(1, C, H * W).contiguous())[0]
        labels, sel = mean_shift_smart_init(X, kappa=20, num_seeds=num_seeds, max_iters=10, metric=metric, precision=precision)
        out_label[j] = labels.view(H, W).float()
        selected_pixels.append(sel)
    return out_label, selected_pixels
