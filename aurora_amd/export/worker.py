"""Drop-in for the reference's ``export/worker.py``: the export subprocess entry point.

Same protocol (export/worker.py:9-85): ``argv[1]`` is a JSON config path; progress is reported as JSON lines
on stdout (``{"type": "progress", "pct": int}``, ``{"type": "done"}``, ``{"type": "error", "message": str}``);
exit code 0 = ok, 1 = export failed, 2 = setup failed.  The config keys are the reference's
(``audio_path, out_path, width, height, fps, mode, color, sensitivity, prefer_hevc?, view_state?, gpu_device?``);
the compute GPU is chosen with ``view_state["hip_device"]`` (``gpu_device`` still names an FFmpeg encoder).
Qt is not needed by the HIP renderer, so no QApplication is created.
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path


def _emit(msg: dict) -> None:
    try:
        sys.stdout.write(json.dumps(msg) + "\n")
        sys.stdout.flush()
    except Exception:
        pass


def main(argv: list[str]) -> int:
    if len(argv) < 2:
        _emit({"type": "error", "message": "Missing config path"})
        return 2
    cfg_path = Path(argv[1]).expanduser().resolve()
    if not cfg_path.exists():
        _emit({"type": "error", "message": f"Config not found: {cfg_path}"})
        return 2
    try:
        if hasattr(os, "nice"):
            os.nice(8)
    except Exception:
        pass
    try:
        cfg = json.loads(cfg_path.read_text(encoding="utf-8"))
    except Exception as e:
        _emit({"type": "error", "message": f"Invalid config JSON: {e}"})
        return 2

    root = cfg.get("project_root")
    if root:
        sys.path.insert(0, str(Path(root).resolve()))

    try:
        try:
            import aurora_amd  # noqa: F401
        except ImportError:      # run as a script: put the repository root (parent of aurora_amd/) on the path
            sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
        from aurora_amd.export.exporter import Exporter
    except Exception as e:
        _emit({"type": "error", "message": f"Failed to import dependencies: {e}"})
        return 2

    try:
        exporter = Exporter(
            audio_path=cfg["audio_path"],
            width=int(cfg["width"]),
            height=int(cfg["height"]),
            fps=int(cfg["fps"]),
            mode=str(cfg["mode"]),
            color=tuple(cfg["color"]),
            sensitivity=float(cfg["sensitivity"]),
            prefer_hevc=bool(cfg.get("prefer_hevc", False)),
            view_state=cfg.get("view_state") or {},
            gpu_device=cfg.get("gpu_device") or "",
        )

        def on_progress(pct: int) -> None:
            _emit({"type": "progress", "pct": int(pct)})

        raw_out = cfg.get("hip_raw_out")       # new, optional: write the rgb24 stream to a file instead of FFmpeg
        if raw_out:
            from aurora_amd.pipeline import FileSink
            sink = FileSink(raw_out)
            try:
                exporter.render_to_sink(sink, progress_cb=on_progress)
            finally:
                sink.close()
            on_progress(100)
        else:
            exporter.render_to_file(cfg["out_path"], progress_cb=on_progress)
        _emit({"type": "done"})
        return 0
    except Exception as e:
        _emit({"type": "error", "message": str(e)})
        return 1


if __name__ == "__main__":
    raise SystemExit(main(sys.argv))
