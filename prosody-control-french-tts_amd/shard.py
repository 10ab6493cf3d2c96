"""Utterance sharding across the GPUs of one node + the single exchange step.

Every measurement is a pure function of one utterance, so ranks own contiguous blocks of
the segment-sorted utterance list (Code/audioPipeline.py:364-367 order) and never talk on
the data path.  Only the baselines (sliding medians, Code/audioPipeline.py:401-424) and the
EMA smoothing (:592-602) need every segment's scalars: one all-gather of fixed-width fp64
records (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  Payload is a
few hundred KB at most, so a single padded all-gather beats anything ring-tuned.  The row counts of every rank
follow from ``shard_range`` (and, for syntagmes, from the TextGrids every rank can read), so nothing but the
records themselves is exchanged.
"""
from __future__ import annotations

import numpy as np

SEGMENT_RECORD = ("p_nat", "l_nat", "l_syn", "d_nat", "d_syn", "wc", "rate_ratio")   # Code/audioPipeline.py:391-400
SYNTAGME_RECORD = ("seg_idx", "p_nat", "l_syn", "nat_total", "syn_total", "pause_ms", "wc_syn")   # measured at :499-523
RECORD_WIDTH = 1 + max(len(SEGMENT_RECORD), len(SYNTAGME_RECORD))   # column 0: 0 = segment row, 1 = syntagme row


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block of ``rank``: gathered order == global (segment) order."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rank_world():
    """(rank, world) of the initialised ``torch.distributed`` process group; (0, 1) when there is none -- the product entry
    points (``AudioPipeline.measure_prosody_and_build_ssml``, ``Aligners.use_whisper_timestamped.main``) shard whenever the
    launcher (``torchrun`` / ``bench.py --gpus N`` style, one process per GPU) has initialised one, and run as before otherwise."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:                                                    # noqa: BLE001  (torch absent: single process)
        pass
    return 0, 1


def exchanging() -> bool:
    """True when the collectives of this module really run: a process group exists and has more than one rank -- or exactly one
    and ``PCE_DIST_WORLD1=1``, the form in which the RCCL code path (device tensors, ``all_gather_into_tensor``, the status
    all-reduce) can execute on a one-GPU box (RCCL refuses two ranks on one device)."""
    import os
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or os.environ.get("PCE_DIST_WORLD1") == "1"
    except Exception:                                                    # noqa: BLE001  (torch absent: single process)
        return False


def local_device() -> int:
    """Device index of this rank under a one-process-per-GPU launcher: ``LOCAL_RANK``, unless ``PCE_RANK_DEVICE`` names the device
    (several ranks sharing ONE GPU -- the only multi-rank form a single-GPU box can run; see :func:`init_from_env`)."""
    import os
    return int(os.environ.get("PCE_RANK_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def init_from_env():
    """Join the launcher's process group (``RANK`` / ``WORLD_SIZE`` / ``MASTER_*`` in the environment) -> (rank, world, device index).
    Backend "nccl" (= RCCL over xGMI), one rank per GPU.  ``PCE_DIST_BACKEND=gloo`` selects gloo for the SAME code path -- the control
    plane and the one all-gather then travel over host memory; RCCL refuses two ranks on one device, so this (with
    ``PCE_RANK_DEVICE=0``) is how the real engine runs under more than one rank on a one-GPU box.  No process group when WORLD_SIZE
    is absent or 1 -- unless ``PCE_DIST_WORLD1=1`` asks for the one-rank group (:func:`exchanging`).  The reference's parallel driver is the spawn pool of Code/audioPipeline.py:1143-1150."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dev = local_device()
    if world > 1 or os.environ.get("PCE_DIST_WORLD1") == "1":
        import torch
        import torch.distributed as dist
        backend = os.environ.get("PCE_DIST_BACKEND", "nccl")
        if backend not in ("nccl", "gloo"):
            raise ValueError(f"PCE_DIST_BACKEND={backend!r}: nccl or gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(dev)
        # A bounded wait in every collective (and in the rendezvous itself): the reference's pool hands back ``(ok, name)`` per voice and
        # ends (Code/audioPipeline.py:1111-1119, 1150-1154); a rank whose peer has died must end too -- with the default (gloo: 30 minutes,
        # RCCL: 10) the survivor sits inside its next status barrier for that long.  The longest legitimate wait is the slowest rank's
        # rank-local section of ONE step (every section is left through ``agreed``), so the bound is generous and adjustable.
        from datetime import timedelta
        timeout = timedelta(seconds=collective_timeout_s())
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev), timeout=timeout)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)
    return rank, world, dev


def collective_timeout_s() -> int:
    """Seconds a rank waits for its peers inside one collective before it raises: ``PCE_DIST_TIMEOUT_S`` (default 300)."""
    import os
    t = int(os.environ.get("PCE_DIST_TIMEOUT_S", "300"))
    if t <= 0:
        raise ValueError(f"PCE_DIST_TIMEOUT_S={t}: a positive number of seconds")
    return t


class PeerFailure(RuntimeError):
    """Another rank failed in the section all ranks have just left (or flagged its block of an exchange as failed): this rank
    stops the same step, so that the next collective every rank enters is the same one."""


def barrier(ok: bool = True) -> bool:
    """Barrier that carries a status: every rank passes whether its rank-local work succeeded and every rank learns whether ALL
    did (one 1-element MIN all-reduce: what ``dist.barrier`` costs, control plane only).  No-op (-> ``ok``) without a process
    group.  A rank must reach this call whether or not its local work raised: see :class:`agreed`."""
    import torch
    import torch.distributed as dist
    if not exchanging():
        return bool(ok)
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


class agreed:
    """``with shard.agreed(): <rank-local work, no collective inside>``: every rank leaves the block through the same status
    barrier whether its work raised or not; a local exception propagates on its rank and every other rank raises
    :class:`PeerFailure`, so a failure on one rank fails the step on all of them together instead of leaving the others inside the
    next collective.  ``only_rank``: the body runs on that rank alone (``if a.mine:``), the others just wait for it."""

    def __init__(self, only_rank=None):
        self.rank = rank_world()[0]
        self.mine = only_rank is None or self.rank == only_rank

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        failed = exc_type is not None and not (exc_type is SystemExit and not getattr(exc, "code", 0))
        all_ok = barrier(not failed)
        if exc_type is None and not all_ok:
            raise PeerFailure("another rank failed in this section")
        return False


def allgather_records(local: np.ndarray, counts=None, device=None, failed: bool = False) -> np.ndarray:
    """All-gather ragged per-rank record blocks ``[n_local, width]`` (float64) into ``[n_total, width]`` in rank order
    with ONE collective (``all_gather_into_tensor``; RCCL over xGMI when the backend is "nccl").

    ``counts``: rows every rank contributes, known to all ranks without talking -- ``shard_range`` sizes for
    per-utterance records, TextGrid-derived syntagme counts for the tagger.  Blocks are padded to ``max(counts)``.
    Without ``counts`` the block sizes have to be agreed on first (one extra scalar MAX all-reduce; the row count then
    rides in a header row): callers on the hot path always pass ``counts``.

    ``failed``: this rank could not produce its block (its measurements raised).  It still takes part in the ONE collective -- the
    block's trailing status row says so -- and every rank, this one included, raises :class:`PeerFailure` after it: a rank-local
    failure never leaves the other ranks waiting inside the exchange, and costs no extra collective."""
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, dtype=np.float64)
    if local.ndim != 2:
        raise ValueError("records must be [n, width]")
    if not exchanging():
        if failed:
            raise PeerFailure("rank 0 could not produce its records")
        if counts is not None and int(counts[0]) != local.shape[0]:
            raise ValueError(f"rank 0 holds {local.shape[0]} records, counts says {counts[0]}")
        return local.copy()
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    width = local.shape[1]
    header = 0
    if counts is None:
        cap = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        cap, header = int(cap.item()), 1
    else:
        counts = [int(c) for c in counts]
        if len(counts) != world:
            raise ValueError(f"counts = {counts} for {world} ranks")          # (the same on every rank: nobody enters the collective)
        if counts[rank] != local.shape[0]:
            failed = True                                                    # a rank-local mismatch must not leave the others waiting
        cap = max(counts) if counts else 0
    if failed:
        local = np.zeros((0, width))
    block = torch.zeros((cap + header + 1, width), dtype=torch.float64, device=device)
    if header:
        block[0, 0] = float(local.shape[0])
    block[cap + header, 0] = 1.0 if failed else 0.0                          # trailing status row
    if local.shape[0]:
        block[header:header + local.shape[0]] = torch.from_numpy(local).to(device)
    out = torch.empty((world * (cap + header + 1), width), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, block)
    out = out.cpu().numpy().reshape(world, cap + header + 1, width)
    bad = [r for r in range(world) if out[r, cap + header, 0] != 0.0]
    if bad:
        raise PeerFailure(f"rank(s) {bad} could not produce their records")
    if header:
        counts = [int(out[r, 0, 0]) for r in range(world)]
    parts = [out[r, header:header + counts[r]] for r in range(world)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, width))
