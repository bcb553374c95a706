"""The reference's own module names (dropin/): importable the way the reference's callers import them, without a GPU.
`ui/main_window.py:17`  from export.exporter import Exporter, build_ffmpeg_cmd, list_gpu_export_devices
`audio/input.py:48`     from audio.analysis import Analyzer
`ui/main_window.py:1138-1151`  python export/worker.py cfg.json  (setup failures: exit 2 + one error line)"""
import inspect
import json
import os
import subprocess
import sys

from conftest import REPO

DROPIN = os.path.join(REPO, "dropin")


def _run(code, cwd):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}      # aurora_amd is NOT on the path: dropin finds it
    return subprocess.run([sys.executable, "-c", code], cwd=cwd, env=env, capture_output=True, text=True, timeout=120)


def test_reference_import_lines_resolve_to_the_hip_path(tmp_path):
    code = (f"import sys; sys.path.insert(0, {DROPIN!r})\n"
            "from export.exporter import Exporter, build_ffmpeg_cmd, list_gpu_export_devices\n"
            "from audio.analysis import Analyzer\n"
            "import inspect, json\n"
            "print(json.dumps({'exporter': Exporter.__module__, 'analyzer': Analyzer.__module__,\n"
            "                  'ctor': list(inspect.signature(Exporter.__init__).parameters),\n"
            "                  'an': list(inspect.signature(Analyzer.__init__).parameters),\n"
            "                  'cmd': list(inspect.signature(build_ffmpeg_cmd).parameters)[:7]}))\n")
    res = _run(code, str(tmp_path))
    assert res.returncode == 0, res.stderr
    got = json.loads(res.stdout.strip().splitlines()[-1])
    assert got["exporter"] == "aurora_amd.export.exporter" and got["analyzer"] == "aurora_amd.audio.analysis"
    # export/exporter.py:495-497, audio/analysis.py:6, export/exporter.py:413-414
    assert got["ctor"] == ["self", "audio_path", "width", "height", "fps", "mode", "color", "sensitivity", "prefer_hevc",
                           "view_state", "gpu_rendering", "gpu_device"]
    assert got["an"][:3] == ["self", "sample_rate", "fft_size"]          # + an optional HIP device index
    assert got["cmd"] == ["width", "height", "fps", "out_path", "audio_path", "prefer_hevc", "gpu_device"]


def test_worker_script_by_its_reference_path_reports_setup_errors(tmp_path):
    worker = os.path.join(DROPIN, "export", "worker.py")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    res = subprocess.run([sys.executable, worker], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 2 and json.loads(res.stdout) == {"type": "error", "message": "Missing config path"}
    bad = tmp_path / "cfg.json"
    bad.write_text("{not json")
    res = subprocess.run([sys.executable, worker, str(bad)], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                         timeout=120)
    assert res.returncode == 2 and json.loads(res.stdout)["message"].startswith("Invalid config JSON")
    # a config the exporter cannot satisfy (no such audio file): the export itself fails -> exit 1 (export/worker.py:77-79)
    cfg = dict(audio_path=str(tmp_path / "missing.wav"), out_path=str(tmp_path / "o.mp4"), width=64, height=64, fps=30,
               mode="Spectrum - Linear", color=[0.2, 0.8, 1.0], sensitivity=1.0)
    ok = tmp_path / "ok.json"
    ok.write_text(json.dumps(cfg))
    res = subprocess.run([sys.executable, worker, str(ok)], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                         timeout=120)
    lines = [json.loads(ln) for ln in res.stdout.strip().splitlines()]
    assert res.returncode == 1 and lines[-1]["type"] == "error"
