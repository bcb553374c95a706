"""The multi-GPU export driver with REAL children on the GPU box: 2-3 worker processes share card 0, each renders its time
shard with the HIP path and writes its own segment; the supervisor (a process that never touches the GPU) joins them.
The join must equal the unsharded stream byte for byte -- "Spectrum - Linear" (carried colour EMA), "Spectrum - Radial"
(carried `_radial_prev` ring state) -- through raw segments and through per-shard (stand-in) FFmpeg + concat."""
import json
import os
import stat
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# stand-in FFmpeg (the binary is not installed): a segment encode copies stdin to the output file, the concat join
# concatenates the listed files -- so the joined "video" is the raw frame stream and can be compared byte for byte
FAKE_FFMPEG = textwrap.dedent('''\
    #!%s
    import re, sys
    a = sys.argv[1:]
    if "-encoders" in a:
        sys.exit(0)
    if a[a.index("-f") + 1] == "concat":
        with open(a[-1], "wb") as out:
            for line in open(a[a.index("-i") + 1]):
                out.write(open(re.match(r"file '(.*)'", line.strip()).group(1), "rb").read())
    else:
        assert "-an" in a and a[a.index("-i") + 1] == "pipe:0", a
        sys.stderr.write("frame= 1 fps=0.0\\n" * 2000)
        with open(a[-1], "wb") as out:
            while True:
                b = sys.stdin.buffer.read(1 << 20)
                if not b:
                    break
                out.write(b)
''') % sys.executable


@pytest.fixture(scope="module")
def eng():
    from aurora_amd.engine import get_engine
    return get_engine(0)


def _write_cfg(tmp_path, mode, raw, vs=None):
    from scipy.io import wavfile
    from aurora_amd import synth
    sr, fps, W, H = 48000, 60, 640, 360
    audio = synth.pink(sr * 3 + 777, 2, seed=31)
    wav = tmp_path / "in.wav"
    wavfile.write(str(wav), sr, audio)
    cfg = dict(audio_path=str(wav), out_path=str(tmp_path / "out.mp4"), width=W, height=H, fps=fps, mode=mode,
               color=[0.2, 0.8, 1.0], sensitivity=1.2, view_state=dict(vs or {}, hip_batch_frames=16), gpu_device="")
    if raw:
        cfg["hip_raw_out"] = str(tmp_path / "frames.rgb")
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    return p, cfg, audio, (sr, fps, W, H)


def _unsharded(eng, audio, dims, mode, sens, vs=None):
    from aurora_amd.engine import RadialConfig
    from aurora_amd.pipeline import FramePipeline
    sr, fps, W, H = dims
    track = eng.upload_track(audio, sr, fps)
    radial = RadialConfig.from_view_state(vs or {}, sens) if "Radial" in mode else None
    return FramePipeline(eng, track, mode, W, H, sens, radial=radial).render_device().cpu().numpy().tobytes()


@pytest.mark.parametrize("mode,world", [("Spectrum - Linear", 3), ("Spectrum - Radial", 2)])
def test_children_on_one_card_join_to_the_unsharded_stream(eng, tmp_path, mode, world):
    vs = {"radial_temporal_smoothing": 60, "radial_mirror": True} if "Radial" in mode else {}
    p, cfg, audio, dims = _write_cfg(tmp_path, mode, raw=True, vs=vs)
    out = subprocess.run([sys.executable, "-m", "aurora_amd.export.sharded", str(p), "--gpus", str(world),
                          "--devices", "0"], cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)
    lines = [json.loads(l) for l in out.stdout.decode().strip().splitlines()]
    assert out.returncode == 0 and lines[-1] == {"type": "done"}, (out.stdout[-2000:], out.stderr[-2000:])
    pcts = [m["pct"] for m in lines if m["type"] == "progress"]
    assert pcts[-1] == 100 and pcts == sorted(pcts)
    got = open(cfg["hip_raw_out"], "rb").read()
    want = _unsharded(eng, audio, dims, mode, 1.2, vs)
    assert len(got) == len(want) and got == want


def test_per_shard_ffmpeg_segments_and_concat(eng, tmp_path, monkeypatch):
    """view_state["hip_gpus"] through the reference's entry point (export.worker): per-shard FFmpeg children (stand-ins)
    and the concat join."""
    ff = tmp_path / "ffmpeg"
    ff.write_text(FAKE_FFMPEG)
    ff.chmod(ff.stat().st_mode | stat.S_IXUSR)
    p, cfg, audio, dims = _write_cfg(tmp_path, "Spectrum - Linear", raw=False, vs={"hip_gpus": 2, "hip_devices": [0]})
    env = dict(os.environ, AURORA_FFMPEG=str(ff))
    out = subprocess.run([sys.executable, "-m", "aurora_amd.export.worker", str(p)], cwd=REPO, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280)
    lines = [json.loads(l) for l in out.stdout.decode().strip().splitlines()]
    assert out.returncode == 0 and lines[-1] == {"type": "done"}, (out.stdout[-2000:], out.stderr[-2000:])
    got = open(cfg["out_path"], "rb").read()
    assert got == _unsharded(eng, audio, dims, "Spectrum - Linear", 1.2)


def test_many_tracks_on_one_lane_with_overlapped_start_up(eng, tmp_path):
    """BASELINE configs[4] in miniature: several tracks dealt to one GPU lane, each exported by its own worker process, the
    next worker started while the current one renders its tail (two workers share the card for a moment).  Every track's
    stream equals its unsharded export."""
    from scipy.io import wavfile
    from aurora_amd import synth
    from aurora_amd.export import sharded
    sr, fps, W, H = 48000, 60, 320, 192
    cfgs, audios = [], []
    for k in range(3):
        audio = synth.pink(sr * 2 + 100 * k, 2, seed=40 + k)
        wav = tmp_path / f"in{k}.wav"
        wavfile.write(str(wav), sr, audio)
        audios.append(audio)
        cfgs.append(dict(audio_path=str(wav), out_path=str(tmp_path / f"o{k}.mp4"), width=W, height=H, fps=fps,
                         mode="Spectrum - Linear", color=[0.2, 0.8, 1.0], sensitivity=1.0,
                         view_state={"hip_batch_frames": 8}, gpu_device="", hip_raw_out=str(tmp_path / f"t{k}.rgb")))
    seen = []
    assert sharded.export_tracks(cfgs, 1, seen.append, devices=[0], prestart_pct=50) == 3
    assert seen == [33, 66, 100]
    for k in range(3):
        got = open(cfgs[k]["hip_raw_out"], "rb").read()
        assert got == _unsharded(eng, audios[k], (sr, fps, W, H), "Spectrum - Linear", 1.0), k


def test_more_shards_than_frames_with_real_workers(eng, tmp_path):
    """A 4-frame track over 6 time shards (more GPUs than frames): the real workers of the empty shards start no FFmpeg and
    leave no segment, the concat join takes the four that exist, and the result is the unsharded stream."""
    from scipy.io import wavfile
    from aurora_amd import synth
    from aurora_amd.export import sharded
    ff = tmp_path / "ffmpeg"
    ff.write_text(FAKE_FFMPEG)
    ff.chmod(ff.stat().st_mode | stat.S_IXUSR)
    sr, fps, W, H = 48000, 60, 320, 192
    audio = synth.pink(800 * 4 - 10, 2, seed=3)
    wav = tmp_path / "short.wav"
    wavfile.write(str(wav), sr, audio)
    cfg = dict(audio_path=str(wav), out_path=str(tmp_path / "short.mp4"), width=W, height=H, fps=fps,
               mode="Spectrum - Linear", color=[0.2, 0.8, 1.0], sensitivity=1.0, view_state={}, gpu_device="")
    old = os.environ.get("AURORA_FFMPEG")
    os.environ["AURORA_FFMPEG"] = str(ff)
    try:
        sharded.export_sharded(cfg, 6, None, devices=[0])
    finally:
        if old is None:
            os.environ.pop("AURORA_FFMPEG", None)
        else:
            os.environ["AURORA_FFMPEG"] = old
    want = _unsharded(eng, audio, (sr, fps, W, H), "Spectrum - Linear", 1.0)
    assert len(want) == 4 * W * H * 3 and open(cfg["out_path"], "rb").read() == want
    assert not list(tmp_path.glob("short.seg*.mp4"))
