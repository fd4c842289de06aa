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


# ---- what the communicator runs over (rank 0 of a multi-GPU bench puts this into its JSON line) -------------------------------
def parse_rccl_debug(text):
    """Summarise an ``NCCL_DEBUG=INFO`` log of ONE rank: which transport each channel connection uses ("via P2P/IPC", "via
    SHM/...", "via NET/..."), whether the topology RCCL detected mentions xGMI links, and the library version line.  Pure text
    processing (tested on CPU with sample logs); the log is produced by bench.py under ``NCCL_DEBUG_FILE``."""
    import re
    via = {}
    for m in re.finditer(r"via\s+([A-Za-z0-9]+)(/[A-Za-z0-9_/]+)?", text):
        key = m.group(1).upper() + (m.group(2) or "")
        via[key] = via.get(key, 0) + 1
    kinds = {k.split("/")[0] for k in via}
    ver = re.search(r"(RCCL|NCCL) version[ :]+([^\s]+)", text)
    xgmi = len(re.findall(r"XGMI", text, flags=re.I))
    if not via:
        transport = "unknown (no channel lines in the log)"
    elif kinds == {"P2P"}:
        transport = "P2P only" + (" (xGMI links in the detected topology)" if xgmi else " (no xGMI link named in the log: PCIe peer access?)")
    else:
        transport = "mixed: " + ", ".join(sorted(kinds))
    return {"transport": transport, "channel_connections": via, "xgmi_mentions": xgmi,
            "library_version_line": ver.group(0) if ver else None}


def communicator_report(dist, debug_file=None):
    """{"backend", "world_size", "rccl_version", + parse_rccl_debug(debug_file)} for the JSON line; never raises."""
    rep = {"backend": None, "world_size": 1}
    try:
        if dist is None or not dist.is_initialized():
            return rep
        rep.update(backend=dist.get_backend(), world_size=dist.get_world_size())
        if rep["backend"] == "nccl":
            try:
                rep["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:                      # noqa: BLE001 -- a report, not the data path
                rep["rccl_version"] = f"unavailable ({e})"
        if debug_file:
            try:
                with open(debug_file, errors="replace") as f:
                    rep.update(parse_rccl_debug(f.read()))
                rep["debug_log"] = debug_file
            except OSError as e:
                rep["transport"] = f"unknown (log {debug_file}: {e})"
    except Exception as e:                              # noqa: BLE001
        rep["error"] = str(e)
    return rep


def timed_all_gather(dist, reps=20):
    """Wall time of the metrics all_gather itself (seconds per call, after one warm-up call): the path's only collective."""
    import time
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0.0
    rec = {k: 0.0 for k in METRIC_KEYS}
    gather_metrics(rec, dist)
    if dist.get_backend() == "nccl":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gather_metrics(rec, dist)
    return (time.perf_counter() - t0) / reps
