"""`export.exporter` of the reference (export/exporter.py:368-567), served by the HIP path.  `ui/main_window.py:17`
imports `Exporter, build_ffmpeg_cmd, list_gpu_export_devices`; `export/worker.py:49` imports `Exporter`."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import _locate  # noqa: E402
_sys.path.pop(0)
_locate.ensure()

from aurora_amd.export.exporter import (  # noqa: E402,F401
    Exporter, HipOffscreenRenderer, build_ffmpeg_cmd, list_gpu_export_devices, load_audio)

# the reference's renderer class name, for code that constructs it directly (export/exporter.py:98)
QPainterOffscreenRenderer = HipOffscreenRenderer

__all__ = ["Exporter", "QPainterOffscreenRenderer", "HipOffscreenRenderer", "build_ffmpeg_cmd",
           "list_gpu_export_devices", "load_audio"]
