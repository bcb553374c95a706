"""aurora_amd: MI355X-native implementation of Aurora's offline-export hot path.

Sub-modules mirror the reference's module names for the path:
``aurora_amd.audio.analysis`` (Analyzer), ``aurora_amd.export.exporter`` (Exporter, renderer, FFmpeg argv),
``aurora_amd.export.worker`` (subprocess protocol).  ``aurora_amd.engine`` / ``aurora_amd.pipeline`` drive the
HIP kernels in ``libaurora_hip.so`` (sources in ``aurora_amd/csrc``, C ABI in ``include/aurora_hip.h``).
"""
__version__ = "0.1.0"
