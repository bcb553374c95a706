"""`export/worker.py` of the reference (export/worker.py:9-85), served by the HIP path: started by the UI as
`python export/worker.py cfg.json` (ui/main_window.py:1138-1151); JSON lines on stdout, exit 0 / 1 / 2."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import _locate  # noqa: E402
_sys.path.pop(0)
_locate.ensure()

from aurora_amd.export.worker import _emit, main  # noqa: E402,F401

if __name__ == "__main__":
    raise SystemExit(main(_sys.argv))
