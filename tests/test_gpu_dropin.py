"""The drop-in by the reference's names on the GPU box: `python dropin/export/worker.py cfg.json` exactly as
`ui/main_window.py:1138-1151` launches it (QT_QPA_PLATFORM=offscreen, cfg in a temporary JSON file, JSON lines read from
stdout as `:1153-1183` does), `from export.exporter import Exporter, ...` (`ui/main_window.py:17`) and
`from audio.analysis import Analyzer` (`audio/input.py:48`) with dropin/ in front of sys.path."""
import hashlib
import json
import os
import stat
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(REPO, "dropin")

# stand-in FFmpeg: stores its stdin at the output path (the binary is not installed on the box)
FAKE_FFMPEG = """#!%s
import shutil, sys
a = sys.argv[1:]
if "-encoders" in a:
    sys.exit(0)
assert a[a.index("-pix_fmt") + 1] == "rgb24" and a[a.index("-i") + 1] == "pipe:0", a
sys.stderr.write("frame= 1 fps=0.0\\n" * 500)
shutil.copyfileobj(sys.stdin.buffer, open(a[-1], "wb"), 1 << 20)
""" % sys.executable


def _env(tmp_path):
    ff = tmp_path / "ffmpeg"
    ff.write_text(FAKE_FFMPEG)
    ff.chmod(ff.stat().st_mode | stat.S_IXUSR)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env.update(AURORA_FFMPEG=str(ff), QT_QPA_PLATFORM="offscreen")
    return env


def _track(tmp_path, sr=48000, seconds=2.5):
    from scipy.io import wavfile
    from aurora_amd import synth
    audio = synth.pink(int(sr * seconds) + 55, 2, seed=23)
    wav = tmp_path / "in.wav"
    wavfile.write(str(wav), sr, audio)
    return audio, wav


@pytest.mark.parametrize("mode", ["Spectrum - Linear", "Waveform - Linear", "Spectrum - Radial"])
def test_worker_launched_as_the_ui_launches_it(tmp_path, mode):
    from oracle import analysis_ref as A
    from oracle import raster_ref as R
    sr, fps, W, H = 48000, 60, 640, 360
    audio, wav = _track(tmp_path, sr)
    out = tmp_path / "out.mp4"
    cfg = dict(project_root=str(tmp_path), audio_path=str(wav), out_path=str(out), width=W, height=H, fps=fps, mode=mode,
               color=[0.2, 0.8, 1.0], sensitivity=1.0, prefer_hevc=False,
               view_state={"hip_stats_path": str(tmp_path / "stats.json")}, gpu_device="")
    cfg_path = tmp_path / "export_cfg.json"
    cfg_path.write_text(json.dumps(cfg))
    res = subprocess.run([sys.executable, os.path.join(DROPIN, "export", "worker.py"), str(cfg_path)], cwd=str(tmp_path),
                         env=_env(tmp_path), capture_output=True, timeout=280)
    lines = [json.loads(ln) for ln in res.stdout.decode().strip().splitlines()]
    assert res.returncode == 0 and lines[-1] == {"type": "done"}, (res.stdout[-2000:], res.stderr[-2000:])
    pcts = [m["pct"] for m in lines if m["type"] == "progress"]
    ref = A.analyze_track(audio, sr, fps)
    F = ref["samples"].shape[0]
    # export/exporter.py:545-549, :561-565: int(i * 100 / total) every max(1, int(fps / 2)) frames, then 100
    assert pcts == [int(i * 100 / F) for i in range(0, F, max(1, int(fps / 2)))] + [100]
    got = np.frombuffer(out.read_bytes(), np.uint8).reshape(F, H, W, 3)
    stats = json.loads((tmp_path / "stats.json").read_text())          # view_state["hip_stats_path"]
    assert stats["frames"] == F and stats["bytes_per_frame"] == W * H * 3 and stats["frame_range"] == [0, F]
    assert stats["fps"] > 0 and stats["pipe"]["spliced_bytes"] + stats["pipe"]["copied_bytes"] == F * W * H * 3
    if mode != "Spectrum - Radial":
        pen, _ = A.amp_to_color_track(ref["energy"])
        for f in (0, 1, F // 2, F - 1):
            want = R.render_frame(mode, ref["samples"][f], ref["logspec"][f], float(ref["energy"][f]), pen[f], W, H)
            assert np.abs(got[f].astype(int) - want.astype(int)).max() <= 1 or (got[f] != want).any(axis=2).sum() <= 4
    else:
        assert (got[0] != got[F // 2]).any() and tuple(got[0, 0, 0]) == (13, 13, 18)


def test_reference_named_imports_render_the_same_stream(tmp_path):
    """ui/main_window.py:17 and audio/input.py:48 import by these names; an Exporter built from them renders the stream
    the package's own Exporter renders, and Analyzer.compute follows the oracle."""
    audio, wav = _track(tmp_path, 44100, 1.5)
    code = (f"import sys, json, hashlib; sys.path.insert(0, {DROPIN!r})\n"
            "from export.exporter import Exporter, build_ffmpeg_cmd, list_gpu_export_devices\n"
            "from audio.analysis import Analyzer\n"
            "import numpy as np\n"
            "class S:\n"
            "    def __init__(s): s.h = hashlib.sha256(); s.n = 0\n"
            "    def write(s, b): s.h.update(b); s.n += len(b)\n"
            f"ex = Exporter({str(wav)!r}, width=640, height=360, fps=30, mode='Spectrum - Linear')\n"
            "s = S(); ex.render_to_sink(s)\n"
            "a = Analyzer(44100, 2048)\n"
            "x = np.random.default_rng(0).standard_normal(1024).astype(np.float32)\n"
            "a.compute(x); spec, flux = a.compute(x)\n"
            "cmd, pix = build_ffmpeg_cmd(640, 360, 30, 'o.mp4', 'a.wav')\n"
            "print(json.dumps({'sha': s.h.hexdigest(), 'n': s.n, 'spec': [float(v) for v in spec[:4]], 'flux': flux,\n"
            "                  'pix': pix, 'devs': list_gpu_export_devices(), 'dtype': str(spec.dtype), 'len': len(spec)}))\n")
    res = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=_env(tmp_path), capture_output=True,
                         text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    got = json.loads(res.stdout.strip().splitlines()[-1])
    from aurora_amd.export.exporter import Exporter

    class Sink:
        def __init__(self):
            self.h, self.n = hashlib.sha256(), 0

        def write(self, b):
            self.h.update(b)
            self.n += len(b)

    s = Sink()
    Exporter(str(wav), width=640, height=360, fps=30, mode="Spectrum - Linear").render_to_sink(s)
    assert got["n"] == s.n and got["sha"] == s.h.hexdigest()
    assert got["pix"] == "rgb24" and got["devs"] == [] and got["dtype"] == "float32" and got["len"] == 1025
    # SURVEY.md 8(c) known answer of the reference's Analyzer: second compute(x) of default_rng(0) noise
    assert np.allclose(got["spec"], [3.939069, 3.587926, 3.061712, 2.5850947], rtol=1e-5)
    assert abs(got["flux"] - 2.37437275152439) < 1e-4
