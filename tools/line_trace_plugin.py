"""pytest plugin (opt-in: `python -m pytest -p tools.line_trace_plugin ...` from the repository root): records which lines of
aurora_amd/*.py the test run executes and writes them to gpurun_out/lines_executed.json.  A development aid for finding
host-side paths no test reaches; spawned worker processes are not followed."""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aurora_amd") + os.sep
_seen = {}


def _local(frame, event, arg):
    if event == "line":
        _seen[frame.f_code.co_filename].add(frame.f_lineno)
    return _local


def _global(frame, event, arg):
    fn = frame.f_code.co_filename
    if not fn.startswith(PKG):
        return None
    _seen.setdefault(fn, set()).add(frame.f_lineno)
    return _local


def pytest_configure(config):
    sys.settrace(_global)
    threading.settrace(_global)


def pytest_unconfigure(config):
    sys.settrace(None)
    out = os.path.join(ROOT, "gpurun_out", "lines_executed.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        json.dump({os.path.relpath(k, ROOT): sorted(v) for k, v in _seen.items()}, fh)
