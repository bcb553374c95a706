"""N > 1 host path on CPU: two gloo ranks shard a track by time, 'render' their ranges with the oracle into per-rank
segment files, exchange digests only, and rank 0 joins the files in rank order into the unsharded byte stream; the job
time is the max over ranks."""
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


def _rank_main(rank, world, port, q, tmpdir):
    import torch.distributed as dist
    from aurora_amd import synth
    from aurora_amd.export import sharded
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
        seg_path = sharded.segment_path(os.path.join(tmpdir, "frames.rgb"), rank)
        with open(seg_path, "wb") as fh:
            for f in range(a, b):
                fh.write(R.render_frame("Spectrum - Linear", ref["samples"][f], ref["logspec"][f],
                                        float(ref["energy"][f]), pen[f], W, H).tobytes())
        recs = sharding.gather_digests_in_order(sharded.file_digest(seg_path), os.path.getsize(seg_path))
        slow = sharding.max_over_ranks(1.0 + rank)
        dist.barrier()                                            # every segment file is complete
        if rank == 0:
            assert [r[0] for r in recs] == list(range(world))
            paths = [sharded.segment_path(os.path.join(tmpdir, "frames.rgb"), r) for r in range(world)]
            joined = os.path.join(tmpdir, "frames.rgb")
            report = sharded.join_raw_segments(paths, joined)
            ok = [(d, n) for _, n, d in report] == [(r[1], r[2]) for r in recs]
            whole = hashlib.sha256()
            for f in range(F):
                whole.update(R.render_frame("Spectrum - Linear", ref["samples"][f], ref["logspec"][f],
                                            float(ref["energy"][f]), pen[f], W, H).tobytes())
            q.put((ok and sharded.file_digest(joined) == whole.hexdigest(), [r[2] for r in recs], slow, F))
    finally:
        dist.destroy_process_group()


def test_two_rank_time_shards_reassemble(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    same, sizes, slow, F = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same
    assert sum(sizes) == F * 160 * 90 * 3 and abs(sizes[0] - sizes[1]) <= 160 * 90 * 3
    assert slow == 2.0
