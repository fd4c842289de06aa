"""Data-parallel plumbing: images are independent units (no cross-image op exists on the path,
pretrained_meanshiftformer_model.py:347-376), so ranks shard the batch and exchange only a small
fixed-size metrics record.  One process per GPU; backend "nccl" is RCCL on ROCm (xGMI), "gloo" on
CPU (tests).  The single collective is a latency-bound all_gather of a few float64 values."""
import torch

METRIC_KEYS = ("images", "elapsed_s", "checksum")


def shard_range(n_items, world_size, rank):
    """Contiguous, balanced [lo, hi) slice of `n_items` for `rank` (first ranks take the remainder)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_metrics(record, dist=None, keys=METRIC_KEYS):
    """all_gather of a per-rank record (dict of numbers) -> list of dicts, one per rank, on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [dict(record)]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([float(record[k]) for k in keys], dtype=torch.float64, device=dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [{k: float(v) for k, v in zip(keys, t.cpu().tolist())} for t in out]
