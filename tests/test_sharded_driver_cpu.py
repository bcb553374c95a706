"""The multi-GPU export driver (aurora_amd/export/sharded.py) on CPU: process supervision, shard configs, ordered join,
progress aggregation, failure handling and the many-tracks lanes, with stand-in worker / FFmpeg child processes (the
real children need a GPU: tests/test_gpu_sharded.py)."""
import json
import os
import stat
import sys
import textwrap

import pytest

from aurora_amd import sharding
from aurora_amd.export import sharded

# A stand-in for `python -m aurora_amd.export.worker cfg.json`: frame f of the track is the byte (f % 251) repeated
# `frame_bytes` times; it writes the frames of its shard (or the whole track) the way the real worker would.
FAKE_WORKER = textwrap.dedent('''
    import json, os, sys, time
    cfg = json.load(open(sys.argv[1]))
    vs = cfg.get("view_state") or {}
    total, fb = int(cfg["fake_total_frames"]), int(cfg["fake_frame_bytes"])
    rank, world = vs.get("hip_shard") or [0, 1]
    base, extra = divmod(total, world)
    a = rank * base + min(rank, extra)
    b = a + base + (1 if rank < extra else 0)
    if cfg.get("fake_fail_rank") == rank:
        print(json.dumps({"type": "error", "message": "boom on rank %d" % rank}), flush=True)
        sys.exit(1)
    if cfg.get("fake_hang_rank") == rank:
        if cfg.get("fake_grandchild"):            # the worker's FFmpeg: must not outlive a stopped worker
            import subprocess
            g = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
            open(cfg["fake_grandchild"], "w").write(str(g.pid))
        time.sleep(600)
    out = cfg.get("hip_raw_out") or cfg["out_path"]
    if a == b and vs.get("hip_segment"):          # the product starts no FFmpeg for a shard without frames
        print(json.dumps({"type": "done"}), flush=True)
        sys.exit(0)
    with open(out, "wb") as fh:
        for f in range(a, b):
            fh.write(bytes([f % 251]) * fb)
            if f % 7 == 0:
                print(json.dumps({"type": "progress", "pct": int(f * 100 / total)}), flush=True)
    t_start = time.time()
    if cfg.get("fake_tail_sleep"):                # the last few per cent of a real export take a while
        print(json.dumps({"type": "progress", "pct": 97}), flush=True)
        time.sleep(float(cfg["fake_tail_sleep"]))
    with open(out + ".meta", "w") as fh:
        json.dump({"device": vs.get("hip_device"), "segment": bool(vs.get("hip_segment")), "pid": os.getpid(),
                   "start": t_start, "end": time.time()}, fh)
    print(json.dumps({"type": "progress", "pct": 100}), flush=True)
    print(json.dumps({"type": "done"}), flush=True)
''')

# A stand-in FFmpeg for the join: `-f concat -i list.txt ... out` concatenates the listed files.
FAKE_FFMPEG = textwrap.dedent('''\
    #!%s
    import re, sys
    a = sys.argv[1:]
    if "-encoders" in a:
        sys.exit(0)
    assert a[a.index("-f") + 1] == "concat", a
    lst = a[a.index("-i") + 1]
    with open(a[-1], "wb") as out:
        for line in open(lst):
            out.write(open(re.match(r"file '(.*)'", line.strip()).group(1), "rb").read())
''') % sys.executable


@pytest.fixture
def fake_worker(tmp_path):
    p = tmp_path / "fake_worker.py"
    p.write_text(FAKE_WORKER)
    return [sys.executable, str(p)]


def _cfg(tmp_path, total=53, fb=16, **extra):
    c = {"audio_path": "a.wav", "out_path": str(tmp_path / "out.mp4"), "width": 4, "height": 2, "fps": 30,
         "mode": "Spectrum - Linear", "color": [0.2, 0.8, 1.0], "sensitivity": 1.0, "view_state": {"hip_gpus": 3},
         "fake_total_frames": total, "fake_frame_bytes": fb}
    c.update(extra)
    return c


def _expected(total, fb):
    return b"".join(bytes([f % 251]) * fb for f in range(total))


def test_shard_configs_are_rank_ordered_ranges(tmp_path):
    cfg = _cfg(tmp_path, hip_raw_out=str(tmp_path / "frames.rgb"))
    cs = sharded.shard_configs(cfg, 4, devices=[0, 1])
    assert [c["view_state"]["hip_shard"] for c in cs] == [[r, 4] for r in range(4)]
    assert [c["view_state"]["hip_device"] for c in cs] == [0, 1, 0, 1]
    assert all("hip_gpus" not in c["view_state"] and "hip_segment" not in c["view_state"] for c in cs)
    assert cs[2]["hip_raw_out"].endswith("frames.seg002.rgb")
    enc = sharded.shard_configs(_cfg(tmp_path), 2)
    assert enc[1]["out_path"].endswith("out.seg001.mp4") and enc[1]["view_state"]["hip_segment"] is True
    assert cfg["view_state"] == {"hip_gpus": 3}                         # the caller's config is not modified
    st = sharded.shard_configs(dict(_cfg(tmp_path), view_state={"hip_stats_path": "/tmp/s.json"}), 2)
    assert [c["view_state"]["hip_stats_path"] for c in st] == ["/tmp/s.seg000.json", "/tmp/s.seg001.json"]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_raw_segments_join_to_the_unsharded_stream(tmp_path, fake_worker, world):
    total, fb = 53, 16
    raw = tmp_path / "frames.rgb"
    seen = []
    report = sharded.export_sharded(_cfg(tmp_path, total, fb, hip_raw_out=str(raw)), world, seen.append,
                                    worker_cmd=fake_worker, devices=[0])
    assert raw.read_bytes() == _expected(total, fb)
    sizes = [n for _, n, _ in report]
    want = [fb * (b - a) for a, b in (sharding.frame_range(total, world, r) for r in range(world))]
    assert sizes == want and len({d for _, _, d in report}) == len(report)
    assert not list(tmp_path.glob("frames.seg*.rgb"))                   # segments removed after the join
    assert seen[-1] == 100 and seen == sorted(seen) and len(set(seen)) == len(seen)


def test_encoded_segments_are_joined_by_ffmpeg_concat(tmp_path, fake_worker, monkeypatch):
    ff = tmp_path / "ffmpeg"
    ff.write_text(FAKE_FFMPEG)
    ff.chmod(ff.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("AURORA_FFMPEG", str(ff))
    cmd = sharded.build_ffmpeg_concat_cmd("l.txt", "a.wav", "o.mp4")
    assert cmd == [str(ff), "-y", "-f", "concat", "-safe", "0", "-i", "l.txt", "-i", "a.wav", "-c:v", "copy",
                   "-c:a", "aac", "-b:a", "192k", "-shortest", "o.mp4"]
    sharded.export_sharded(_cfg(tmp_path, 40, 8), 3, None, worker_cmd=fake_worker, devices=[0], keep_segments=True)
    assert (tmp_path / "out.mp4").read_bytes() == _expected(40, 8)
    metas = [json.load(open(str(tmp_path / f"out.seg{r:03d}.mp4") + ".meta")) for r in range(3)]
    assert all(m["segment"] for m in metas) and len({m["pid"] for m in metas}) == 3   # one process per shard


def test_segment_command_is_the_reference_encode_without_audio(tmp_path, monkeypatch):
    from aurora_amd.export import exporter
    ff = tmp_path / "ffmpeg"
    ff.write_text(FAKE_FFMPEG)
    ff.chmod(ff.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("AURORA_FFMPEG", str(ff))
    full, _ = exporter.build_ffmpeg_cmd(3840, 2160, 60, "o.mp4", "a.wav")
    seg, _ = exporter.build_ffmpeg_segment_cmd(3840, 2160, 60, "o.seg000.mp4")
    i = full.index("a.wav")
    assert seg[:i - 1] == full[:i - 1]                                   # same raw-video input
    j = full.index("-c:a")
    assert seg[i - 1:-2] == full[i + 1:j] and seg[-2:] == ["-an", "o.seg000.mp4"]   # same encoder arguments


def test_a_failing_shard_stops_the_others(tmp_path, fake_worker):
    cfg = _cfg(tmp_path, hip_raw_out=str(tmp_path / "frames.rgb"), fake_fail_rank=1, fake_hang_rank=2)
    with pytest.raises(RuntimeError) as ei:
        sharded.export_sharded(cfg, 3, None, worker_cmd=fake_worker, devices=[0])
    assert "boom on rank 1" in str(ei.value)
    assert not (tmp_path / "frames.rgb").exists()
    assert not list(tmp_path.glob("frames.seg*.rgb"))                   # rank 0 finished its segment: not left behind


def test_stopping_a_worker_stops_what_it_started(tmp_path, fake_worker):
    """A failing shard ends the others AND their children (a worker's FFmpeg must not go on writing a segment the
    supervisor removes): workers run in their own process group and the group is signalled."""
    import time
    pidfile = tmp_path / "grandchild.pid"
    cfg = _cfg(tmp_path, hip_raw_out=str(tmp_path / "frames.rgb"), fake_fail_rank=1, fake_hang_rank=2,
               fake_grandchild=str(pidfile))

    # rank 1 fails only after rank 2 has started its grandchild
    def slow_fail_worker():
        p = tmp_path / "slow_fail_worker.py"
        p.write_text("import json, os, sys, time\ncfg = json.load(open(sys.argv[1]))\n"
                     "r = (cfg.get('view_state') or {}).get('hip_shard', [0, 1])[0]\n"
                     "if r == 1:\n"
                     "    t0 = time.time()\n"
                     "    while not os.path.exists(cfg['fake_grandchild']) and time.time() - t0 < 30: time.sleep(0.05)\n"
                     "    time.sleep(0.2)\n"
                     f"exec(open({str(tmp_path / 'fake_worker.py')!r}).read())\n")
        return [sys.executable, str(p)]

    with pytest.raises(RuntimeError):
        sharded.export_sharded(cfg, 3, None, worker_cmd=slow_fail_worker(), devices=[0])
    pid = int(pidfile.read_text())
    for _ in range(100):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        # a zombie re-parented to init still answers kill(0) until it is reaped: look at its state
        try:
            if open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] == "Z":
                break
        except OSError:
            break
        time.sleep(0.05)
    else:
        os.kill(pid, 9)
        raise AssertionError("the stopped worker's child is still running")


def test_a_silent_worker_is_stopped_after_the_stall_timeout(tmp_path, fake_worker):
    """Opt-in stall detection: a worker that says nothing for `hip_stall_timeout_s` seconds is taken for hung; the export
    fails with a message that names it instead of waiting for good."""
    import time
    cfg = _cfg(tmp_path, hip_raw_out=str(tmp_path / "frames.rgb"), fake_hang_rank=1)
    cfg["view_state"] = {"hip_stall_timeout_s": 1.0}
    t0 = time.time()
    with pytest.raises(RuntimeError) as ei:
        sharded.export_sharded(cfg, 2, None, worker_cmd=fake_worker, devices=[0])
    assert "shard001: no sign of life for 1 s" in str(ei.value) and time.time() - t0 < 30
    assert not list(tmp_path.glob("frames.seg*.rgb")) and not (tmp_path / "frames.rgb").exists()


def test_a_worker_that_ignores_sigterm_is_killed(tmp_path, fake_worker, monkeypatch):
    """A failing shard stops the others with SIGTERM; one that does not react (blocked in a call) gets SIGKILL after a
    grace period, so the supervisor's waiting threads always return.  A malformed progress line does not end the reader."""
    import time
    monkeypatch.setattr(sharded._Child, "KILL_AFTER_S", 0.5)
    p = tmp_path / "deaf_worker.py"
    p.write_text("import json, signal, sys, time\ncfg = json.load(open(sys.argv[1]))\n"
                 "r = (cfg.get('view_state') or {}).get('hip_shard', [0, 1])[0]\n"
                 "if r == 0:\n"
                 "    signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
                 "    print(json.dumps({'type': 'progress', 'pct': 'soon'}), flush=True)\n"
                 "    print(json.dumps({'type': 'progress', 'pct': 3}), flush=True)\n"
                 "    time.sleep(600)\n"
                 "time.sleep(0.5)\n"
                 "print(json.dumps({'type': 'error', 'message': 'boom'}), flush=True)\nsys.exit(1)\n")
    seen = []
    t0 = time.time()
    with pytest.raises(RuntimeError) as ei:
        sharded.export_sharded(_cfg(tmp_path, hip_raw_out=str(tmp_path / "frames.rgb")), 2, seen.append,
                               worker_cmd=[sys.executable, str(p)], devices=[0])
    assert "boom" in str(ei.value) and time.time() - t0 < 20
    assert seen == [3]                  # the line after the malformed one was still read


def test_more_gpus_than_frames(tmp_path, fake_worker, monkeypatch):
    """Shards with an empty frame range: a raw segment of zero bytes joins as nothing; an encoded shard starts no FFmpeg
    and leaves no segment, and the concat join takes the segments that exist."""
    raw = tmp_path / "frames.rgb"
    sharded.export_sharded(_cfg(tmp_path, 3, 8, hip_raw_out=str(raw)), 8, None, worker_cmd=fake_worker, devices=[0])
    assert raw.read_bytes() == _expected(3, 8)
    ff = tmp_path / "ffmpeg"
    ff.write_text(FAKE_FFMPEG)
    ff.chmod(ff.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("AURORA_FFMPEG", str(ff))
    sharded.export_sharded(_cfg(tmp_path, 3, 8), 8, None, worker_cmd=fake_worker, devices=[0])
    assert (tmp_path / "out.mp4").read_bytes() == _expected(3, 8)


def test_many_tracks_are_dealt_over_gpu_lanes(tmp_path, fake_worker):
    cfgs = [_cfg(tmp_path, 10 + k, 4, hip_raw_out=str(tmp_path / f"t{k}.rgb")) for k in range(7)]
    seen = []
    assert sharded.export_tracks(cfgs, 3, seen.append, worker_cmd=fake_worker) == 7
    for k in range(7):
        assert (tmp_path / f"t{k}.rgb").read_bytes() == _expected(10 + k, 4)
        assert json.load(open(str(tmp_path / f"t{k}.rgb") + ".meta"))["device"] == k % 3     # track k on GPU k mod 3
    assert seen[-1] == 100


def test_next_tracks_worker_starts_during_the_tail_of_the_current_one(tmp_path, fake_worker):
    """A lane starts the next track's worker once the current one reports 96 % (start-up hidden behind the tail); with
    prestart_pct=101 the workers of a lane run strictly one after another."""
    def run(prestart, tag):
        cfgs = [_cfg(tmp_path, 10, 4, hip_raw_out=str(tmp_path / f"{tag}{k}.rgb"), fake_tail_sleep=0.6) for k in range(3)]
        assert sharded.export_tracks(cfgs, 1, None, worker_cmd=fake_worker, prestart_pct=prestart) == 3
        return [json.load(open(str(tmp_path / f"{tag}{k}.rgb") + ".meta")) for k in range(3)]
    over = run(96, "a")
    assert all((tmp_path / f"a{k}.rgb").read_bytes() == _expected(10, 4) for k in range(3))
    assert over[1]["start"] < over[0]["end"]                                             # overlapped start-up
    assert over[2]["start"] >= over[0]["end"] - 0.05                                     # never three at once
    serial = run(101, "b")
    assert serial[1]["start"] >= serial[0]["end"] and serial[2]["start"] >= serial[1]["end"]


def test_cli_and_worker_dispatch(tmp_path, fake_worker, capsys, monkeypatch):
    cfg = _cfg(tmp_path, hip_raw_out=str(tmp_path / "frames.rgb"))
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    monkeypatch.setattr(sharded, "default_worker_cmd", lambda: fake_worker)
    assert sharded.main(["sharded", str(p), "--gpus", "2", "--devices", "0"]) == 0
    lines = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    assert lines[-1] == {"type": "done"} and lines[-2] == {"type": "progress", "pct": 100}
    assert (tmp_path / "frames.rgb").read_bytes() == _expected(53, 16)
    # the reference's entry point with view_state["hip_gpus"] > 1 supervises instead of rendering
    from aurora_amd.export import worker
    os.remove(tmp_path / "frames.rgb")
    assert worker.main(["worker.py", str(p)]) == 0
    assert (tmp_path / "frames.rgb").read_bytes() == _expected(53, 16)
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1]) == {"type": "done"}
    assert sharded.main(["sharded", str(tmp_path / "missing.json"), "--gpus", "2"]) == 2


def test_numa_placement_helpers(tmp_path):
    """sharding.bind_near_device: reads the GPU's NUMA node and local CPUs from sysfs (a stand-in tree here), binds only to
    CPUs the process may use, and leaves a host that hides its topology alone."""
    import os
    from aurora_amd import sharding
    assert sharding.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert sharding.parse_cpulist("") == set()
    dev = tmp_path / "0000:c1:00.0"
    dev.mkdir()
    (dev / "numa_node").write_text("1\n")
    allowed = sorted(os.sched_getaffinity(0))
    (dev / "local_cpulist").write_text(f"{allowed[0]}\n")
    assert sharding.device_numa("0000:C1:00.0", str(tmp_path)) == (1, {allowed[0]})
    assert sharding.device_numa("0000:ff:00.0", str(tmp_path)) == (None, set())

    class _Lib:
        def __init__(self, bus):
            self.bus = bus

        def aur_device_pci_bus_id(self, device, buf, n):
            if self.bus is None:
                return -1
            buf.value = self.bus
            return 0

    import threading
    before = os.sched_getaffinity(0)
    stop = threading.Event()
    helper = threading.Thread(target=stop.wait, daemon=True)
    helper.start()
    try:
        info = sharding.bind_near_device(_Lib(b"0000:c1:00.0"), 3, str(tmp_path))
        assert {k: info[k] for k in ("device", "bus_id", "numa_node", "bound_cpus")} == \
            {"device": 3, "bus_id": "0000:c1:00.0", "numa_node": 1, "bound_cpus": 1}
        assert os.sched_getaffinity(0) == {allowed[0]}
        if len(allowed) > 1:
            # threads that existed before the call are bound too (HIP / RCCL helper threads in a real rank)
            assert info["bound_threads"] >= 2 and os.sched_getaffinity(helper.native_id) == {allowed[0]}
    finally:
        os.sched_setaffinity(0, before)
        sharding._bind_other_threads(before)
        stop.set()
    # the same without any library: the KFD topology names the GPU's PCI address (no HIP call before the binding)
    kfd = tmp_path / "kfd"
    for i, (simd, loc) in enumerate([(0, 0), (1024, 0x0d00), (1024, 0xc100)]):
        (kfd / str(i)).mkdir(parents=True)
        (kfd / str(i) / "properties").write_text(f"cpu_cores_count 8\nsimd_count {simd}\nlocation_id {loc}\ndomain 0\n")
    assert sharding.pci_bus_id_from_sysfs(1, str(kfd), {}) == "0000:c1:00.0"
    assert sharding.pci_bus_id_from_sysfs(0, str(kfd), {"ROCR_VISIBLE_DEVICES": "1"}) == "0000:c1:00.0"
    assert sharding.pci_bus_id_from_sysfs(0, str(kfd), {"HIP_VISIBLE_DEVICES": "1", "CUDA_VISIBLE_DEVICES": "0"}) == "0000:c1:00.0"
    assert sharding.pci_bus_id_from_sysfs(2, str(kfd), {}) is None
    assert sharding.pci_bus_id_from_sysfs(0, str(kfd), {"ROCR_VISIBLE_DEVICES": "GPU-1234"}) is None
    assert sharding.pci_bus_id_from_sysfs(0, str(tmp_path / "nokfd"), {}) is None
    hidden = {k: os.environ.pop(k) for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")
              if k in os.environ}
    try:
        info = sharding.bind_near_device(None, 1, str(tmp_path), str(kfd))
        assert info["bus_id"] == "0000:c1:00.0" and info["via"] == "sysfs" and info["bound_cpus"] == 1
        assert os.sched_getaffinity(0) == {allowed[0]}
    finally:
        os.sched_setaffinity(0, before)
        os.environ.update(hidden)
    assert sharding.bind_near_device(None, 0, str(tmp_path), str(tmp_path / "nokfd"))["bus_id"] is None
    (dev / "numa_node").write_text("-1\n")                      # virtual machines
    assert sharding.bind_near_device(_Lib(b"0000:c1:00.0"), 0, str(tmp_path))["bound_cpus"] == 0
    assert os.sched_getaffinity(0) == before
    assert sharding.bind_near_device(_Lib(None), 0, str(tmp_path))["bus_id"] is None
    (dev / "numa_node").write_text("0\n")
    (dev / "local_cpulist").write_text("100000-100003\n")       # CPUs this process may not use: nothing changes
    assert sharding.bind_near_device(_Lib(b"0000:c1:00.0"), 0, str(tmp_path))["bound_cpus"] == 0
    assert os.sched_getaffinity(0) == before
