"""Puts the repository that holds the `aurora_amd` package on sys.path (AURORA_AMD_ROOT, else the parent of dropin/)."""
import os
import sys


def ensure():
    try:
        import aurora_amd  # noqa: F401  (already importable: installed, or on PYTHONPATH)
        return
    except ImportError:
        pass
    root = os.environ.get("AURORA_AMD_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
