"""Utterance sharding across the GPUs of one node + the single exchange step.

Every measurement is a pure function of one utterance, so ranks own contiguous blocks of
the segment-sorted utterance list (Code/audioPipeline.py:364-367 order) and never talk on
the data path.  Only the baselines (sliding medians, Code/audioPipeline.py:401-424) and the
EMA smoothing (:592-602) need every segment's scalars: one all-gather of fixed-width fp64
records (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  Payload is a
few hundred KB at most, so a single padded all-gather beats anything ring-tuned.
"""
from __future__ import annotations

import numpy as np

SEGMENT_RECORD = ("p_nat", "l_nat", "l_syn", "d_nat", "d_syn", "wc", "rate_ratio")   # Code/audioPipeline.py:391-400


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block of ``rank``: gathered order == global (segment) order."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_records(local: np.ndarray, device=None) -> np.ndarray:
    """All-gather ragged per-rank record blocks ``[n_local, width]`` (float64) into
    ``[n_total, width]`` in rank order.  One collective for the counts is avoided by padding
    to the maximum block size carried in the record header row."""
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, dtype=np.float64)
    if local.ndim != 2:
        raise ValueError("records must be [n, width]")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.copy()
    world = dist.get_world_size()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    width = local.shape[1]
    # counts ride along in one extra header row, so a single collective moves everything
    cap = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    cap = int(cap.item())
    block = torch.zeros((cap + 1, width), dtype=torch.float64, device=device)
    block[0, 0] = float(local.shape[0])
    if local.shape[0]:
        block[1:1 + local.shape[0]] = torch.from_numpy(local).to(device)
    out = torch.empty((world * (cap + 1), width), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, block)
    out = out.cpu().numpy().reshape(world, cap + 1, width)
    parts = [out[r, 1:1 + int(out[r, 0, 0])] for r in range(world)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, width))
