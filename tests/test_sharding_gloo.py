"""N > 1 host path on CPU: two gloo ranks shard a track by time, 'render' their ranges with the oracle and the
final gather reassembles the unsharded byte stream; the job time is the max over ranks."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from aurora_amd import sharding


def test_frame_ranges_tile_the_track():
    for total in (0, 1, 7, 900, 14400, 216000):
        for world in (1, 2, 3, 4, 8):
            cuts = [sharding.frame_range(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.track_assignment(64, 8, 3) == list(range(3, 64, 8))
    with pytest.raises(ValueError):
        sharding.frame_range(10, 2, 2)


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    from aurora_amd import synth
    from oracle import analysis_ref as A
    from oracle import raster_ref as R
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sr, fps, W, H = 48000, 60, 160, 90
        audio = synth.pink(sr // 2 + 77, 2, seed=3)
        ref = A.analyze_track(audio, sr, fps)                     # sequential state: every rank needs the prefix
        pen, _ = A.amp_to_color_track(ref["energy"])
        F = ref["samples"].shape[0]
        a, b = sharding.frame_range(F, world, rank)
        seg = b"".join(R.render_frame("Spectrum - Linear", ref["samples"][f], ref["logspec"][f],
                                      float(ref["energy"][f]), pen[f], W, H).tobytes() for f in range(a, b))
        parts = sharding.gather_segments_in_order(seg)
        slow = sharding.max_over_ranks(1.0 + rank)
        if rank == 0:
            whole = b"".join(R.render_frame("Spectrum - Linear", ref["samples"][f], ref["logspec"][f],
                                            float(ref["energy"][f]), pen[f], W, H).tobytes() for f in range(F))
            q.put((hashlib.sha256(b"".join(parts)).hexdigest() == hashlib.sha256(whole).hexdigest(),
                   [len(p) for p in parts], slow, F))
    finally:
        dist.destroy_process_group()


def test_two_rank_time_shards_reassemble():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, sizes, slow, F = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same
    assert sum(sizes) == F * 160 * 90 * 3 and abs(sizes[0] - sizes[1]) <= 160 * 90 * 3
    assert slow == 2.0
