"""Time sharding of an export across the GPUs of one node (one process per GPU, no data-path collective).

The reference renders one frame after another in a single process (export/exporter.py:538-549).  Frames only
depend on the past through the analysis halo (previous block / previous magnitudes, rebuilt inside the STFT
kernel) and three scalar recurrences (carried state, ``HipExportEngine.carried_state``), so contiguous frame
ranges are independent units of work: rank r renders ``frame_range(total, world, r)`` and the host writes the
segments to the sink in rank order.  Whole tracks (BASELINE.json configs[4]: 64 tracks on 8 GPUs) are dealt
round-robin with ``track_assignment``.
"""
from __future__ import annotations


def frame_range(total_frames: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced [begin, end) of ``rank``; ranges of all ranks tile [0, total_frames) in rank order."""
    total_frames, world, rank = int(total_frames), int(world), int(rank)
    if world < 1 or not (0 <= rank < world) or total_frames < 0:
        raise ValueError(f"bad shard request: total={total_frames} world={world} rank={rank}")
    base, extra = divmod(total_frames, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def track_assignment(n_tracks: int, world: int, rank: int) -> list[int]:
    """Indices of the tracks ``rank`` exports when whole tracks are dealt round-robin."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad shard request: world={world} rank={rank}")
    return list(range(int(rank), int(n_tracks), int(world)))


def max_over_ranks(seconds: float, device=None) -> float:
    """Slowest rank's time (the job's time).  Uses the default process group if one is initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digests_in_order(digest: str, n_bytes: int, dst: int = 0):
    """The only exchange of a time-sharded export between ranks that share a process group: every rank writes its segment
    to its own file (or its own FFmpeg) and hands ``dst`` a (rank, sha256 hex digest, byte count) record -- never the
    frames themselves -- so ``dst`` can verify the segments it joins in rank order = frame order.  Returns the records
    on ``dst`` and None elsewhere.  (``aurora_amd.export.sharded`` needs no process group at all: its children are
    independent processes and the supervisor reads the segment files.)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [(0, str(digest), int(n_bytes))]
    rec = (dist.get_rank(), str(digest), int(n_bytes))
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(rec, out, dst=dst)
    return out
