"""ctypes binding of libaurora_hip.so (include/aurora_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, this module
raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C aurora_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# AURORA_HIP_LIB selects another build of the same ABI (kernel-tuning experiments); default is the in-tree library
LIB_PATH = os.environ.get("AURORA_HIP_LIB") or os.path.join(_HERE, "libaurora_hip.so")

ABI_VERSION = 1

# every symbol include/aurora_hip.h declares (tests check the library exports exactly these)
SYMBOLS = (
    "aur_version", "aur_last_error", "aur_device_count", "aur_device_pci_bus_id", "aur_create", "aur_destroy", "aur_get_window",
    "aur_stft_frames", "aur_mix_channels", "aur_stft_buffer", "aur_log1p_f32", "aur_scan_frames",
    "aur_spans_spectrum_linear", "aur_spans_particles", "aur_spans_waveform_linear", "aur_raster_spans",
    "aur_raster_work_bytes", "aur_span_summary", "aur_raster_spans_ws", "aur_raster_spans_static",
    "aur_radial_stage1", "aur_radial_ema", "aur_radial_vertices", "aur_envelope_vertices",
    "aur_feather_ring_vertices", "aur_blend_planes", "aur_blend_planes_box", "aur_mask_polygon_box", "aur_add_planes", "aur_multiply_plane", "aur_plane_exclude", "aur_wave_circular_vertices", "aur_mask_stroke_closed", "aur_mask_stroke_path", "aur_wave_linear_vertices", "aur_mask_polygon", "aur_raster_planes", "aur_raster_closed_path", "aur_disc_rows",
    "aur_rgb24_to_yuv420p", "aur_raster_spans_yuv420p", "aur_rgb24_to_nv12", "aur_raster_spans_nv12",
    "aur_glow_work_bytes", "aur_glow_path",
    "aur_image_scale", "aur_center_image_placement", "aur_disc_composite", "aur_blend_pixels", "aur_compose_background", "aur_set_background",
    "aur_host_alloc", "aur_host_free", "aur_readback_async", "aur_stream_synchronize",
)

K_SPECTRUM, K_PARTICLES, K_WAVEFORM = 2, 2, 5
FFT_SIZE, BLOCK, NBINS = 2048, 1024, 1025
STATIC_WORK_BYTES = 3 * 1024 * 1024     # AUR_STATIC_WORK_BYTES


class AuroraHipError(RuntimeError):
    pass


class ScanParams(C.Structure):
    """struct aur_scan_params (include/aurora_hip.h). Defaults = RTVisualizerWidget/Analyzer defaults."""
    _fields_ = [
        ("sensitivity", C.c_double),
        ("amp_alpha", C.c_double),
        ("grad_min", C.c_double),
        ("grad_max", C.c_double),
        ("grad_curve", C.c_int),
        ("grad_a", C.c_uint8 * 4),
        ("grad_b", C.c_uint8 * 4),
        ("flux_alpha", C.c_double),
        ("amp_ema0", C.c_double),
        ("flux_smooth0", C.c_double),
        ("phase0", C.c_double),
    ]

    CURVES = {"linear": 0, "ease-in": 1, "ease-out": 2, "smoothstep": 3}

    @classmethod
    def defaults(cls, sensitivity=1.0):
        p = cls()
        p.sensitivity = max(0.05, float(sensitivity))     # widgets/rt_widget.py:142
        p.amp_alpha = 0.2                                  # rt_widget.py:57
        p.grad_min, p.grad_max = 0.0, 1.0                  # rt_widget.py:54-55
        p.grad_curve = 0                                   # rt_widget.py:53
        p.grad_a[:] = (32, 255, 255, 0)                    # rt_widget.py:51
        p.grad_b[:] = (255, 0, 255, 0)                     # rt_widget.py:52
        p.flux_alpha = 0.9                                 # audio/analysis.py:12
        p.amp_ema0 = p.flux_smooth0 = p.phase0 = 0.0
        return p


class EnvelopeParams(C.Structure):
    """struct aur_envelope_params (include/aurora_hip.h)."""
    _fields_ = [
        ("radial_mirror", C.c_int),
        ("edge_waviness", C.c_int),
        ("feather_audio_enabled", C.c_int),
        ("feather_audio_amount", C.c_int),
        ("feather_sensitivity", C.c_double),
        ("waveform_sensitivity", C.c_double),
        ("rotation_rad", C.c_double),
    ]


class DiscSource(C.Structure):
    """struct aur_disc_source (include/aurora_hip.h)."""
    _fields_ = [
        ("mode", C.c_int32),
        ("colour_argb_pm", C.c_uint32),
        ("d_image", C.c_void_p),
        ("img_w", C.c_int32),
        ("img_h", C.c_int32),
        ("d_placement", C.c_void_p),
        ("placement_static", C.c_int32),
    ]


_lock = threading.Lock()
_lib = None


def _declare(lib):
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.aur_version.restype = i32
    lib.aur_version.argtypes = []
    lib.aur_last_error.restype = C.c_char_p
    lib.aur_last_error.argtypes = []
    lib.aur_device_count.argtypes = [C.POINTER(i32)]
    lib.aur_device_pci_bus_id.argtypes = [i32, C.c_char_p, i32]
    lib.aur_create.argtypes = [i32, C.POINTER(vp)]
    lib.aur_destroy.restype = None
    lib.aur_destroy.argtypes = [vp]
    lib.aur_get_window.argtypes = [vp, vp]
    lib.aur_stft_frames.argtypes = [vp, vp, i64, i32, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.aur_mix_channels.argtypes = [vp, vp, i64, i32, vp, vp]
    lib.aur_stft_buffer.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.aur_log1p_f32.argtypes = [vp, vp, vp, i64, vp]
    lib.aur_scan_frames.argtypes = [vp, vp, vp, i32, C.POINTER(ScanParams), vp, vp, vp, vp, vp, vp]
    lib.aur_spans_spectrum_linear.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    lib.aur_spans_particles.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.aur_spans_waveform_linear.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.aur_raster_spans.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp]
    lib.aur_raster_work_bytes.argtypes = [i32, i32, C.POINTER(C.c_size_t)]
    lib.aur_span_summary.argtypes = [vp, vp, i32, i32, i32, i32, vp, C.c_size_t, vp]
    lib.aur_raster_spans_ws.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp, C.c_size_t, i32, vp]
    lib.aur_raster_spans_static.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp]
    f64 = C.c_double
    lib.aur_radial_stage1.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp]
    lib.aur_radial_ema.argtypes = [vp, vp, i32, i32, f64, i32, vp, vp]
    lib.aur_radial_vertices.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, i32, vp, i32, vp]
    lib.aur_envelope_vertices.argtypes = [vp, vp, vp, vp, i32, C.POINTER(EnvelopeParams), vp, i32, i32, vp, vp]
    lib.aur_feather_ring_vertices.argtypes = [vp, vp, vp, vp, i32, C.POINTER(EnvelopeParams), vp, i32, i32, vp, vp]
    lib.aur_plane_exclude.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, i32, vp]
    lib.aur_blend_planes.argtypes = [vp, vp, i32, C.c_char_p, i32, i32, i32, vp, vp]
    lib.aur_blend_planes_box.argtypes = [vp, vp, i32, C.c_char_p, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.aur_multiply_plane.argtypes = [vp, vp, C.c_char_p, i32, i32, i32, vp, vp]
    lib.aur_add_planes.argtypes = [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.aur_wave_circular_vertices.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, vp]
    lib.aur_mask_stroke_closed.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.aur_rgb24_to_yuv420p.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.aur_raster_spans_yuv420p.argtypes = [vp, vp, i32, vp, i32, i32, i32, vp, vp]
    lib.aur_rgb24_to_nv12.argtypes = lib.aur_rgb24_to_yuv420p.argtypes
    lib.aur_raster_spans_nv12.argtypes = lib.aur_raster_spans_yuv420p.argtypes
    lib.aur_mask_stroke_path.argtypes = [vp, vp, i32, i32, i32, i32, C.c_double, C.c_double, C.c_double, i32, i32, vp, vp]
    lib.aur_wave_linear_vertices.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.aur_mask_polygon.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.aur_mask_polygon_box.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.aur_raster_planes.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, vp, vp]
    lib.aur_raster_closed_path.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp]
    lib.aur_disc_rows.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.aur_glow_work_bytes.argtypes = [i32, i32, C.POINTER(C.c_size_t)]
    lib.aur_glow_path.argtypes = [vp, vp, i32, i32, i32, i32, i32, C.POINTER(C.c_double), vp, vp, vp, i32, vp,
                                  C.c_size_t, vp, vp]
    lib.aur_image_scale.argtypes = [vp, vp, i32, i32, vp, i32, i32, vp]
    lib.aur_center_image_placement.argtypes = [vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.aur_disc_composite.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(DiscSource), i32,
                                       vp, vp, vp]
    lib.aur_blend_pixels.argtypes = [vp, vp, vp, i32, C.c_uint32, i32, i32, i32, vp, vp]
    lib.aur_compose_background.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.aur_set_background.argtypes = [vp, vp, i32, i32]
    lib.aur_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.aur_host_free.argtypes = [vp]
    lib.aur_readback_async.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.aur_stream_synchronize.argtypes = [vp, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("aur_last_error", "aur_destroy"):
            fn.restype = i32


def load():
    """Load (once) and return the ctypes library.  Raises AuroraHipError if it is not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise AuroraHipError(
                    f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()); "
                    "aurora_amd has no CPU fallback")
            # torch first: its bundled HIP/HSA runtime must be the one this library binds to.  Loaded the other
            # way round the process holds two HSA runtimes and the second finds no device (hipGetDeviceCount fails).
            import torch  # noqa: F401
            lib = C.CDLL(LIB_PATH)
            missing = [s for s in SYMBOLS if not hasattr(lib, s)]
            if missing:
                raise AuroraHipError(f"{LIB_PATH} lacks symbols {missing}")
            _declare(lib)
            if lib.aur_version() != ABI_VERSION:
                raise AuroraHipError(f"ABI version {lib.aur_version()} != expected {ABI_VERSION}")
            _lib = lib
        return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().aur_last_error().decode("utf-8", "replace")
        raise AuroraHipError(f"{what or 'aurora_hip call'} failed (code {rc}): {msg}")


class Context:
    """Owns one aur_ctx (one per device per thread)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.device = int(device)
        h = C.c_void_p()
        check(self.lib.aur_create(self.device, C.byref(h)), "aur_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.aur_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def window(self):
        import numpy as np
        w = np.empty(FFT_SIZE, dtype=np.float32)
        check(self.lib.aur_get_window(self.handle, w.ctypes.data_as(C.c_void_p)), "aur_get_window")
        return w
