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


# ---- `python bench.py --gpus N` without a launcher: the parent starts its own ranks (sharding.launch_ranks) ------------

_STANDIN = r"""
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = sys.argv[1]
open(os.path.join(sys.argv[2], f"rank{rank}.json"), "w").write(json.dumps(
    {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                    "HSA_ENABLE_IPC_MODE_LEGACY")} | {"argv": sys.argv[1:]}))
if mode == "ok":
    print(json.dumps({"n_gpus": world, "from_rank": rank}), flush=True)   # every rank prints: only rank 0's is stdout
elif mode == "fail" and rank == 1:
    sys.exit(7)
elif mode == "fail":
    time.sleep(120)                                                       # a rank waiting in a barrier for good
"""


def _run_launcher(tmp_path, mode, world):
    import subprocess
    import sys
    script = tmp_path / "standin_rank.py"
    script.write_text(_STANDIN)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from aurora_amd.sharding import launch_ranks; "
            "sys.exit(launch_ranks(%d, [%r, %r], %r))" % (root, world, mode, str(tmp_path), str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=100, env=env)


def test_launcher_starts_one_rank_per_gpu_and_forwards_rank0(tmp_path):
    import json
    res = _run_launcher(tmp_path, "ok", 3)
    assert res.returncode == 0, res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert [json.loads(ln) for ln in lines] == [{"n_gpus": 3, "from_rank": 0}]       # ONE line, rank 0's
    assert sorted(json.loads(ln)["from_rank"] for ln in res.stderr.splitlines() if ln.startswith("{")) == [1, 2]
    envs = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert {e["WORLD_SIZE"] for e in envs} == {"3"} and {e["MASTER_ADDR"] for e in envs} == {"127.0.0.1"}
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and envs[0]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert envs[1]["argv"][0] == "ok"


def test_launcher_failed_rank_ends_the_job_with_its_status(tmp_path):
    import time
    t0 = time.time()
    res = _run_launcher(tmp_path, "fail", 2)
    assert res.returncode == 7
    assert time.time() - t0 < 60          # rank 0 was terminated, not waited for


def test_bench_parent_touches_no_gpu_before_launching(tmp_path, monkeypatch):
    # bench.py --gpus 2 with no WORLD_SIZE must reach launch_ranks without importing torch
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1']\n"
            "import aurora_amd.sharding as s\n"
            "def fake(world, argv, script, **kw):\n"
            "    assert 'torch' not in sys.modules, 'parent imported torch before starting its ranks'\n"
            "    print('LAUNCH', world, argv, script.endswith('bench.py')); return 0\n"
            "s.launch_ranks = fake\n"
            "runpy.run_path(%r, run_name='__main__')\n" % os.path.join(root, "bench.py"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["PYTHONPATH"] = root
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=100, env=env)
    assert res.returncode == 0, res.stderr
    assert "LAUNCH 2 ['--gpus', '2', '--steps', '1'] True" in res.stdout
