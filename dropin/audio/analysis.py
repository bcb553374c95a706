"""`audio.analysis` of the reference (audio/analysis.py:5-33), served by the HIP path: same class name, constructor and
`compute(x) -> (float32[1025], float)`; imported by `audio/input.py:48` and `export/exporter.py:114`."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import _locate  # noqa: E402
_sys.path.pop(0)
_locate.ensure()

from aurora_amd.audio.analysis import Analyzer  # noqa: E402,F401

__all__ = ["Analyzer"]
