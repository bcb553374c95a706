"""Export subprocess entry point on MI355X, written from the protocol the UI speaks, which the
reference implements in ``export/worker.py:9-85`` and consumes in ``ui/main_window.py:1138-1228``:

* invocation  ``python worker.py <cfg.json>``;
* config keys ``project_root?, audio_path, out_path, width, height, fps, mode, color[3], sensitivity, prefer_hevc?,
  view_state?, gpu_device?``  (``gpu_device`` names an FFmpeg *encoder*; the compute GPU is ``view_state["hip_device"]``);
* stdout      one JSON object per line: ``{"type": "progress", "pct": int}``, ``{"type": "done"}``,
  ``{"type": "error", "message": str}``;
* exit status 0 = exported, 1 = the export itself failed, 2 = could not get as far as exporting (arguments, config,
  imports).

Additions: ``hip_raw_out`` (write the raw frame stream to a file instead of FFmpeg), ``view_state["hip_stats_path"]`` (the
export's frames / seconds / frames per second / wire GB/s as a JSON file) and ``view_state["hip_gpus"] = N``
(N > 1: this process only supervises; ``aurora_amd.export.sharded`` starts one child per GPU -- HIP devices
``view_state["hip_devices"]``, default 0 .. N-1 -- and joins the segments).
Qt is not needed by the HIP renderer, so no QApplication is created.
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path

EXIT_OK, EXIT_EXPORT_FAILED, EXIT_SETUP_FAILED = 0, 1, 2

# config key -> (Exporter keyword, conversion, required)
_EXPORTER_ARGS = (
    ("audio_path", "audio_path", lambda v: v, True),
    ("width", "width", int, True),
    ("height", "height", int, True),
    ("fps", "fps", int, True),
    ("mode", "mode", str, True),
    ("color", "color", tuple, True),
    ("sensitivity", "sensitivity", float, True),
    ("prefer_hevc", "prefer_hevc", bool, False),
    ("view_state", "view_state", lambda v: v or {}, False),
    ("gpu_device", "gpu_device", lambda v: v or "", False),
)
_OPTIONAL_DEFAULTS = {"prefer_hevc": False, "view_state": {}, "gpu_device": ""}


class _Stop(Exception):
    """Carries an exit status and the message for the UI out of any depth of the worker."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def _emit(msg: dict) -> None:
    """One protocol message = one JSON line on stdout, flushed; a closed stdout must not kill the export."""
    try:
        print(json.dumps(msg), file=sys.stdout, flush=True)
    except Exception:
        pass


def _load_cfg(argv) -> dict:
    if len(argv) < 2:
        raise _Stop(EXIT_SETUP_FAILED, "Missing config path")
    path = Path(argv[1]).expanduser().resolve()
    if not path.exists():
        raise _Stop(EXIT_SETUP_FAILED, f"Config not found: {path}")
    try:
        cfg = json.loads(path.read_text(encoding="utf-8"))
    except Exception as e:
        raise _Stop(EXIT_SETUP_FAILED, f"Invalid config JSON: {e}")
    if not isinstance(cfg, dict):
        raise _Stop(EXIT_SETUP_FAILED, "Invalid config JSON: not an object")
    return cfg


def _be_nice() -> None:
    """The export runs next to an interactive UI: lower priority, like the reference's worker (nice 8)."""
    nice = getattr(os, "nice", None)
    if nice is not None:
        try:
            nice(8)
        except OSError:
            pass


def _import_exporter(cfg):
    root = cfg.get("project_root")
    if root:
        sys.path.insert(0, str(Path(root).resolve()))
    try:
        try:
            import aurora_amd  # noqa: F401
        except ImportError:      # run as a script: put the repository root (parent of aurora_amd/) on the path
            sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
        from aurora_amd.export import exporter
    except Exception as e:
        raise _Stop(EXIT_SETUP_FAILED, f"Failed to import dependencies: {e}")
    return exporter


def exporter_kwargs(cfg: dict) -> dict:
    """Exporter keyword arguments from a worker config (a missing required key is a KeyError = an export failure,
    exit 1, as in the reference)."""
    kw = {}
    for key, name, conv, required in _EXPORTER_ARGS:
        if required or key in cfg:
            kw[name] = conv(cfg[key])
        else:
            kw[name] = conv(_OPTIONAL_DEFAULTS[key])
    return kw


def _write_stats(path, stats) -> None:
    """view_state["hip_stats_path"]: the export's own figures (frames, seconds, frames/s, wire GB/s, how the pipe was fed)
    as one JSON object; best effort, never the reason an export fails."""
    if not path or stats is None:
        return
    try:
        Path(path).write_text(json.dumps(stats), encoding="utf-8")
    except OSError:
        pass


def _export(cfg, exporter_mod) -> None:
    def progress(pct):
        _emit({"type": "progress", "pct": int(pct)})

    vs = cfg.get("view_state") or {}
    gpus = int(vs.get("hip_gpus", 1) or 1)
    if gpus > 1:                                 # supervise one child per GPU; this process never touches a GPU
        from aurora_amd.export import sharded
        sharded.export_sharded(cfg, gpus, progress, devices=vs.get("hip_devices"))   # default: devices 0 .. gpus-1
        return
    if "hip_device" in vs and vs.get("hip_bind_numa", True):
        # one of several workers on a multi-GPU node: keep this process (and the pinned frame ring it is about to
        # allocate) on the socket its GPU hangs off
        from aurora_amd import sharding
        dev = int(vs.get("hip_device") or 0)
        if not sharding.bind_near_device(None, dev)["bus_id"]:        # sysfs first: no HIP call before the binding
            from aurora_amd import _lib
            sharding.bind_near_device(_lib.load(), dev)
    ex = exporter_mod.Exporter(**exporter_kwargs(cfg))
    raw_out = cfg.get("hip_raw_out")
    try:
        if not raw_out:
            ex.render_to_file(cfg["out_path"], progress_cb=progress)
            return
        from aurora_amd.pipeline import FileSink
        sink = FileSink(raw_out)
        try:
            ex.render_to_sink(sink, progress_cb=progress)
        finally:
            sink.close()
        progress(100)
    finally:
        _write_stats(vs.get("hip_stats_path"), getattr(ex, "last_stats", None))


def main(argv: list[str]) -> int:
    try:
        cfg = _load_cfg(argv)
        _be_nice()
        exporter_mod = _import_exporter(cfg)
        try:
            _export(cfg, exporter_mod)
        except _Stop:
            raise
        except Exception as e:
            raise _Stop(EXIT_EXPORT_FAILED, str(e))
    except _Stop as stop:
        _emit({"type": "error", "message": str(stop)})
        return stop.status
    _emit({"type": "done"})
    return EXIT_OK


if __name__ == "__main__":
    raise SystemExit(main(sys.argv))
