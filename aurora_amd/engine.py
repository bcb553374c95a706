"""Host-side driver of the MI355X export path: track upload, frame plan, analysis -> scan -> spans -> fill.

Python mirrors of the reference's per-frame glue live here:

* ``frame_plan``        <- export/exporter.py:229-230 (hop, pos) and :113/:536 (total_frames)
* ``waveform_col2samp`` <- widgets/rt_widget.py:908 (column -> sample index map)
* ``mode_kind``         <- widgets/rt_widget.py:840-849 (mode dispatch)

Index maps are computed on the host in float64 with the reference's expression order (they must be
bit-exact) and uploaded; everything per-sample / per-pixel runs in the HIP kernels behind the C ABI
(include/aurora_hip.h).  torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import os

import numpy as np

from . import _lib

NBINS = _lib.NBINS
BLOCK = _lib.BLOCK


def frame_plan(n_samples: int, sample_rate: int, fps: int):
    """(hop, total_frames, pos[int64]) exactly as the exporter computes them, vectorised in float64."""
    sr, fps = int(sample_rate), int(fps)
    hop = max(1, int(sr / fps))                                   # exporter.py:229
    duration = float(n_samples / sr)                              # exporter.py:113
    total = int(np.ceil(duration * fps))                          # exporter.py:536
    t = np.arange(total, dtype=np.float64) / np.float64(fps)      # t = i / fps          (:539)
    x = t * np.float64(sr)                                        # t * sr               (:230)
    x = np.maximum(0.0, np.minimum(np.float64(n_samples - 1), x))
    return hop, total, x.astype(np.int64)                         # int() truncation


def waveform_col2samp(w: int, n: int = BLOCK) -> np.ndarray:
    """``int((x / max(1, w - 1)) * (n - 1))`` for every column x (rt_widget.py:908), float64."""
    x = np.arange(w, dtype=np.float64)
    return ((x / np.float64(max(1, w - 1))) * np.float64(n - 1)).astype(np.int32)


def mode_kind(mode: str) -> str:
    if mode.startswith("Waveform"):
        return "waveform_circular" if "Circular" in mode else "waveform_linear"
    if mode.startswith("Spectrum"):
        return "spectrum_radial" if "Radial" in mode else "spectrum_linear"
    return "particles"


SUPPORTED_KINDS = ("spectrum_linear", "waveform_linear", "particles", "spectrum_radial", "waveform_circular")
PLANE_KINDS = ("spectrum_radial", "waveform_circular")       # rendered through bit planes, not column spans


@dataclass
class RadialConfig:
    """Settings of "Spectrum - Radial" as an export sees them: widget defaults (rt_widget.py:59-67, :106-114)
    overridden by the exporter's _apply_view_state (export/exporter.py:153-209)."""
    radial_rotation_deg: float = 0.0
    radial_mirror: bool = False
    radial_smooth_amount: int = 50
    radial_wave_smoothness: int = 50
    radial_temporal_alpha: float = 0.3
    edge_waviness: int = 0
    feather_enabled: bool = False
    feather_audio_enabled: bool = False
    feather_audio_amount: int = 0
    feather_sensitivity: float = 2.0
    waveform_sensitivity: float = 1.0
    center_image_path: str | None = None     # rt_widget.py:226-240; None / unreadable file: the (30, 30, 40) fill
    center_image_zoom: int = 100             # rt_widget.py:248-254
    center_motion: int = 0                   # rt_widget.py:256-262

    @classmethod
    def from_view_state(cls, vs, sensitivity=1.0):
        vs = vs or {}
        c = cls()
        c.radial_rotation_deg = float(vs.get("radial_rotation_deg", 0.0))
        c.radial_mirror = bool(vs.get("radial_mirror", False))
        c.radial_smooth_amount = max(0, min(100, int(vs.get("radial_smooth_amount", 50))))
        if "radial_waveform_smoothness" in vs:
            c.radial_wave_smoothness = max(0, min(100, int(vs["radial_waveform_smoothness"])))
        if "radial_temporal_smoothing" in vs:
            c.radial_temporal_alpha = max(0.0, min(0.95, int(vs["radial_temporal_smoothing"]) / 100.0))
        c.edge_waviness = max(0, min(100, int(vs.get("edge_waviness", 0))))
        c.feather_enabled = bool(vs.get("feather_enabled", False))
        c.feather_audio_enabled = bool(vs.get("feather_audio_enabled", False))
        c.feather_audio_amount = max(0, min(100, int(vs.get("feather_audio_amount", 0))))
        c.waveform_sensitivity = max(0.05, float(sensitivity))
        c.center_motion = max(0, min(100, int(vs.get("center_motion", 0))))            # exporter.py:164
        if "center_image_zoom" in vs:
            c.center_image_zoom = max(50, min(250, int(vs.get("center_image_zoom", 100))))   # exporter.py:165-166
        c.center_image_path = vs.get("center_image_path") or None                      # exporter.py:171-177
        return c

    @property
    def disc_static(self):
        """The envelope is the same circle in every frame (rt_widget.py:1038 reduces to r = inner)."""
        return self.edge_waviness == 0 and not (self.feather_audio_enabled and self.feather_audio_amount > 0)

    # ring lengths after mirror / _smooth_closed #1 / _smooth_closed #2 (rt_widget.py:946-962, :1125)
    @property
    def L0(self):
        return 158 if self.radial_mirror else 80

    @property
    def L1(self):
        return self.L0 * (1 + self.radial_wave_smoothness // 25 if self.radial_wave_smoothness > 0 else 1)

    @property
    def L2(self):
        return self.L1 * (1 + self.radial_smooth_amount // 25 if self.radial_smooth_amount > 0 else 1)


def _div255(x: int) -> int:
    return (x + (x >> 8) + 0x80) >> 8


@dataclass
class FillConfig:
    """`_apply_fill` (rt_widget.py:856-864) as an export configures it (export/exporter.py:188-191): the closed path of
    "Spectrum - Radial" / "Waveform - Circular" is filled before it is stroked when energy >= threshold."""
    enabled: bool = False
    rgba: tuple = (255, 255, 255, 128)       # '#80FFFFFF', exporter.py:189
    blend: str = "normal"                    # 'normal' | 'add' | 'multiply', rt_widget.py:859-862
    threshold: float = 0.1

    @staticmethod
    def parse_colour(text, fallback):
        """QColor(str) for the '#' forms Qt accepts -- '#AARRGGBB' (what the UI writes: `name(QColor.HexArgb)`,
        export/exporter.py:27), '#RRGGBB', '#RGB', '#RRRGGGBBB', '#RRRRGGGGBBBB' (the high byte of each channel) -- as
        (r, g, b, a); anything else is an invalid QColor and keeps the previous colour (rt_widget.py:369-372)."""
        t = str(text).strip()
        if not t.startswith("#"):
            return fallback
        t = t[1:]
        try:
            if len(t) == 8:
                a, r, g, b = (int(t[i:i + 2], 16) for i in (0, 2, 4, 6))
                return (r, g, b, a)
            if len(t) in (3, 6, 9, 12):
                n = len(t) // 3
                r, g, b = (int((t[i * n:(i + 1) * n] * 2)[:2], 16) for i in range(3))     # '#RGB': F -> FF
                return (r, g, b, 255)
        except ValueError:
            pass
        return fallback

    @classmethod
    def from_view_state(cls, vs):
        vs = vs or {}
        c = cls()
        c.enabled = bool(vs.get("radial_fill_enabled", False))
        c.rgba = cls.parse_colour(vs.get("radial_fill_color", "#80FFFFFF"), (255, 255, 255, 48))   # rt_widget.py:102
        c.blend = vs.get("radial_fill_blend", "normal") or "normal"
        try:
            c.threshold = float(max(0.0, min(1.0, float(vs.get("radial_fill_threshold", 0.1)))))
        except (TypeError, ValueError):
            c.threshold = 0.1
        return c

    def colour_over_background(self, bg=(13, 13, 18)):
        """The fill is painted straight onto the background (paint_frame fills the frame first, rt_widget.py:829), so
        its result is one colour: Qt's premultiplied ARGB32 SourceOver / Plus / Multiply of the fill over opaque bg."""
        r, g, b, a = (int(v) for v in self.rgba)
        out = []
        for c, d in zip((r, g, b), bg):
            s = _div255(c * a)                                   # qPremultiply
            if self.blend == "add":
                v = min(255, s + d)                              # comp_func_Plus
            elif self.blend == "multiply":
                v = _div255(s * d + d * (255 - a))               # multiply_op with an opaque destination
            else:
                v = s + _div255(d * (255 - a))                   # comp_func_SourceOver
            out.append(v)
        return tuple(out)


@dataclass
class ShadowConfig:
    """`_draw_shadow_path` (rt_widget.py:579-594) as an export configures it (export/exporter.py:212-223; widget
    defaults rt_widget.py:80-85): the closed paths ("Spectrum - Radial", "Waveform - Circular") and the linear waveform's
    own variant."""
    enabled: bool = False
    opacity: float = 0.6
    blur: int = 16
    distance: int = 8
    angle_deg: int = 45

    @classmethod
    def from_view_state(cls, vs):
        vs = vs or {}
        c = cls()
        if "shadow_enabled" in vs:
            c.enabled = bool(vs["shadow_enabled"])
        if "shadow_opacity" in vs:
            c.opacity = max(0.0, min(1.0, int(vs["shadow_opacity"]) / 100.0))          # rt_widget.py:302-304
        if "shadow_blur_radius" in vs:
            c.blur = int(max(0, int(vs["shadow_blur_radius"])))
        if "shadow_distance" in vs:
            c.distance = int(max(0, int(vs["shadow_distance"])))
        if "shadow_angle_deg" in vs:
            c.angle_deg = int(vs["shadow_angle_deg"])
        return c

    @property
    def active(self):
        return self.enabled and self.opacity > 0.0                                       # rt_widget.py:580

    @property
    def alpha(self):
        return int(255 * self.opacity)                                                   # :582

    @property
    def offset(self):
        import math
        ang, dist = float(self.angle_deg), float(max(0.0, self.distance))
        return dist * math.cos(math.radians(ang)), dist * math.sin(math.radians(ang))    # :584-587

    @property
    def half_width(self):
        return max(2, self.blur) / 2.0                                                   # pen.setWidth(max(2, blur)), :591

    # "Waveform - Linear" draws its own shadow (rt_widget.py:915-928): RoundCap / RoundJoin pen of width
    # max(2, min(10, blur)), translated by (distance, distance) whatever the angle
    @property
    def linear_pen_width(self):
        return max(2, min(10, self.blur))                                                # :918

    @property
    def linear_offset(self):
        d = float(self.distance)                                                         # :924-925
        return d, d


@dataclass
class GlowConfig:
    """`_draw_glow_path` / `_draw_glow_offscreen` (rt_widget.py:598-616, :634-653) as an export configures it
    (export/exporter.py:196-199; the export always takes the offscreen branch, exporter.py:118): the closed path of
    "Spectrum - Radial" / "Waveform - Circular" stroked `layers` times, CompositionMode_Plus, before its own stroke."""
    enabled: bool = False
    rgba: tuple = (0xA0, 0xFF, 0xFF, 0x80)   # '#80A0FFFF', exporter.py:197
    radius: int = 22
    strength: float = 0.8
    layers: int = 8                          # rt_widget.py:91; no setter, no view_state key

    @classmethod
    def from_view_state(cls, vs):
        vs = vs or {}
        c = cls()
        c.enabled = bool(vs.get("glow_enabled", False))
        # an invalid string keeps the widget's own default (rt_widget.py:88, :343-346)
        c.rgba = FillConfig.parse_colour(vs.get("glow_color", "#80A0FFFF"), (80, 220, 255, 255))
        c.radius = int(max(0, int(vs.get("glow_radius", 22))))                           # :349-351
        c.strength = max(0.0, min(1.0, float(int(vs.get("glow_strength", 80))) / 100.0)) # :356-358
        return c

    @property
    def active(self):
        return self.enabled and self.radius > 0 and self.strength > 0.0                  # :599-604

    @property
    def base_alpha(self):
        """QColor.alphaF(): Qt 6 keeps 16 bits per channel (a * 0x101) and returns alpha / 65535 as a C float."""
        a = np.float32(int(self.rgba[3]) * 257) / np.float32(65535)
        return max(0.0, min(1.0, float(a)))                                              # :612

    def layer_table(self):
        """[(pen width, falloff)] of the layers that can paint (falloff > 0), in paint order (:635-648)."""
        n = max(4, min(16, int(self.layers)))
        out = []
        for i in range(n):
            t = i / float(max(1, n - 1))
            falloff = (1.0 - t) ** 2
            if falloff > 0.0:
                out.append((max(3, int(2 + self.radius * (0.35 + 0.9 * t))), falloff))
        return out

    def layer_alpha(self, energy: np.ndarray) -> np.ndarray:
        """Pen alpha int64 [F, K] of every layer from the frames' energy (float64 [F]), the reference's expression
        order (:606-607, :642); 0 = the layer is skipped in that frame (:643-644)."""
        e = np.asarray(energy, dtype=np.float64)
        intensity = self.strength * (0.15 + 0.85 * np.maximum(0.0, np.minimum(1.0, e)))
        ba = self.base_alpha
        cols = [np.trunc(255 * ba * intensity * 0.55 * falloff) for _, falloff in self.layer_table()]
        return np.maximum(np.stack(cols, axis=1), 0.0).astype(np.int64)

    def layer_premultiplied(self, energy: np.ndarray) -> np.ndarray:
        """uint8 [F, K, 4]: qPremultiply of the pen colour of every layer, (div255(r a), div255(g a), div255(b a), 0)."""
        a = self.layer_alpha(energy)
        out = np.zeros(a.shape + (4,), dtype=np.uint8)
        for ch in range(3):
            x = int(self.rgba[ch]) * a
            out[..., ch] = (x + (x >> 8) + 0x80) >> 8
        return out


def angle_table(n: int, rotation_deg: float) -> np.ndarray:
    """(cos a_i, sin a_i), a_i = 2 pi (i / n) - pi / 2 + deg2rad(rot), i = 0..n, float64 [n + 1, 2], evaluated with the
    reference's expression and scalar calls (rt_widget.py:966, :1035) -- an index-map-like table, so it is exact."""
    out = np.empty((n + 1, 2), dtype=np.float64)
    for i in range(n + 1):
        a = 2.0 * np.pi * (i / n) - (np.pi / 2.0) + np.deg2rad(rotation_deg)
        out[i, 0] = np.cos(a)
        out[i, 1] = np.sin(a)
    return out


def wave_circular_angle_table(rotation_deg: float, n: int = BLOCK) -> np.ndarray:
    """(cos a_i, sin a_i), a_i = 2 pi (i / max(1, n - 1)) - pi / 2 + deg2rad(rot), i < n (rt_widget.py:887), float64
    [n, 2] with the reference's expression and scalar calls."""
    out = np.empty((n, 2), dtype=np.float64)
    for i in range(n):
        a = 2.0 * np.pi * (i / max(1, n - 1)) - (np.pi / 2.0) + np.deg2rad(rotation_deg)
        out[i, 0] = np.cos(a)
        out[i, 1] = np.sin(a)
    return out


@dataclass
class RadialPrep:
    cfg: RadialConfig
    vals1: "torch.Tensor"        # float64 [F, L1]: ring after stage 1 and the temporal EMA
    state: "torch.Tensor"        # float64 [L1]: EMA state after the last frame
    cos_sin: "torch.Tensor"      # float64 [L2 + 1, 2]
    cos_sin360: "torch.Tensor"   # float64 [361, 2]
    static_disc: dict = None     # (W, H) -> what a frame-independent disc needs (HipExportEngine._static_disc)
    image: "torch.Tensor | None" = None   # centre image, premultiplied 0xAARRGGBB int32 [h, w] on the device


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


@dataclass
class Track:
    audio: "torch.Tensor"        # float32 [n, C] on device
    pos: "torch.Tensor"          # int64 [total] on device
    pos_host: np.ndarray
    n_samples: int
    channels: int
    sample_rate: int
    fps: int
    hop: int
    total_frames: int

    @property
    def duration(self) -> float:
        return float(self.n_samples / self.sample_rate)


@dataclass
class Analysis:
    frame_begin: int
    n_frames: int
    logspec: "torch.Tensor | None"   # float32 [F, 1025]
    flux_sum: "torch.Tensor"     # float32 [F]
    meanabs: "torch.Tensor"      # float32 [F]
    samples: "torch.Tensor | None"   # float32 [F, 1024]
    bandmax: "torch.Tensor | None"   # float32 [F, 2]


@dataclass
class Scan:
    energy: "torch.Tensor"       # float64 [F]
    pen_rgb: "torch.Tensor"      # uint8 [F, 4]
    flux_smooth: "torch.Tensor"  # float64 [F]
    phase: "torch.Tensor"        # float64 [F]
    state: "torch.Tensor"        # float64 [3] = (amp_ema, flux_smooth, phase) after the last frame


class HipExportEngine:
    """One engine per device.  All launches go to torch's current stream on that device."""

    def __init__(self, device: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise _lib.AuroraHipError("no HIP device visible: aurora_amd needs an MI355X (no CPU fallback)")
        self.torch = torch
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.ctx = _lib.Context(self.device_index)
        self.lib = self.ctx.lib
        self._tables = {}
        self._work = None
        self._glow_work = None
        self._base = None
        self.fill_events = None           # benchmarks set a list: (start, end, n_frames) HIP events around every plane fill
        import os
        self.direct_stroke = os.environ.get("AUR_DIRECT_STROKE", "1") != "0"   # A/B knob: 0 = always through the stroke plane
        try:                                         # what aur_create read into ctx->fill_mode (csrc/abi.hip)
            self.fill_mode = int(os.environ.get("AUR_FILL_MODE", "0") or 0)
        except ValueError:
            self.fill_mode = 0

    # ------------------------------------------------------------------ plumbing
    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ track
    def make_track(self, audio_dev, sample_rate: int, fps: int) -> Track:
        """Wrap audio already resident in HBM (float32 [n] or [n, C])."""
        torch = self.torch
        if audio_dev.dim() == 1:
            audio_dev = audio_dev.unsqueeze(1)
        if audio_dev.dtype != torch.float32 or not audio_dev.is_contiguous():
            audio_dev = audio_dev.to(torch.float32).contiguous()
        n, ch = int(audio_dev.shape[0]), int(audio_dev.shape[1])
        if ch > 2:
            # more than two channels (5.1 / 7.1 files): one mono mix per track in numpy's y.mean(axis=1) order
            # (exporter.py:233-234), then the mono path -- the samples every frame sees are bit-identical
            if ch > 64:
                raise _lib.AuroraHipError(f"{ch}-channel audio unsupported (at most 64 channels)")
            mono = self.empty((max(n, 1), 1), torch.float32)
            _lib.check(self.lib.aur_mix_channels(self.ctx.handle, _ptr(audio_dev), n, ch, _ptr(mono), self.stream()),
                       "aur_mix_channels")
            audio_dev, ch = mono[:n], 1
        hop, total, pos = frame_plan(n, sample_rate, fps)
        if n < 2:       # the kernel loads sample pairs: a 0- or 1-sample file gets zero rows, which is what the
            audio_dev = torch.cat([audio_dev, audio_dev.new_zeros((2 - n, ch))], dim=0)   # reference's padding reads
        pos_dev = torch.from_numpy(pos).to(self.device)
        return Track(audio_dev, pos_dev, pos, n, ch, int(sample_rate), int(fps), hop, total)

    def upload_track(self, audio: np.ndarray, sample_rate: int, fps: int) -> Track:
        a = np.ascontiguousarray(audio, dtype=np.float32)
        return self.make_track(self.torch.from_numpy(a).to(self.device), sample_rate, fps)

    # ------------------------------------------------------------------ analysis
    def analyze(self, track: Track, frame_begin: int = 0, n_frames: int | None = None, want_samples=False,
                want_bandmax=False, want_logspec=True, reuse: "Analysis | None" = None) -> Analysis:
        """``reuse``: an Analysis of the same shape whose tensors are overwritten instead of allocating new ones (callers
        that analyse the same range again and again; nothing of the old contents survives)."""
        torch = self.torch
        if n_frames is None:
            n_frames = track.total_frames - frame_begin
        if frame_begin < 0 or n_frames < 0 or frame_begin + n_frames > track.total_frames:
            raise ValueError(f"frame range [{frame_begin}, {frame_begin + n_frames}) outside track "
                             f"of {track.total_frames} frames")
        F = int(n_frames)
        if (reuse is not None and reuse.n_frames == F and (reuse.logspec is not None) == bool(want_logspec)
                and (reuse.samples is not None) == bool(want_samples) and (reuse.bandmax is not None) == bool(want_bandmax)):
            logspec, flux_sum, meanabs, samples, bandmax = (reuse.logspec, reuse.flux_sum, reuse.meanabs, reuse.samples,
                                                            reuse.bandmax)
        else:
            logspec = self.empty((F, NBINS), torch.float32) if want_logspec else None
            flux_sum = self.empty((F,), torch.float32)
            meanabs = self.empty((F,), torch.float32)
            samples = self.empty((F, BLOCK), torch.float32) if want_samples else None
            bandmax = self.empty((F, 2), torch.float32) if want_bandmax else None
        _lib.check(self.lib.aur_stft_frames(
            self.ctx.handle, _ptr(track.audio), max(2, track.n_samples), track.channels, _ptr(track.pos),
            int(frame_begin), F, track.hop, _ptr(logspec), _ptr(flux_sum), _ptr(meanabs), _ptr(samples),
            _ptr(bandmax), self.stream()), "aur_stft_frames")
        return Analysis(int(frame_begin), F, logspec, flux_sum, meanabs, samples, bandmax)

    def scan(self, an: Analysis, params: _lib.ScanParams | None = None, state=None) -> Scan:
        """state: None (track start) or (amp_ema, flux_smooth, phase) carried from the previous shard."""
        torch = self.torch
        # a private copy: the carried state of THIS call never leaks into the caller's (reusable) parameter object
        p = _lib.ScanParams.defaults() if params is None else _lib.ScanParams.from_buffer_copy(params)
        p.amp_ema0, p.flux_smooth0, p.phase0 = (float(v) for v in state) if state is not None else (0.0, 0.0, 0.0)
        F = an.n_frames
        energy = self.empty((F,), torch.float64)
        pen = self.empty((F, 4), torch.uint8)
        fs = self.empty((F,), torch.float64)
        ph = self.empty((F,), torch.float64)
        st = self.empty((3,), torch.float64)
        _lib.check(self.lib.aur_scan_frames(
            self.ctx.handle, _ptr(an.meanabs), _ptr(an.flux_sum), F, C.byref(p), _ptr(energy), _ptr(pen),
            _ptr(fs), _ptr(ph), _ptr(st), self.stream()), "aur_scan_frames")
        return Scan(energy, pen, fs, ph, st)

    def carried_state(self, track: Track, frame_begin: int, params: _lib.ScanParams | None = None,
                      chunk: int = 32768):
        """Exact smoothing state in front of ``frame_begin``: analysis-only pre-pass over frames [0, frame_begin).

        The pre-pass moves ~10 KB per frame (no rasterisation), i.e. < 0.1 % of rendering the same frames at 4K.
        """
        state = (0.0, 0.0, 0.0)
        f = 0
        while f < frame_begin:
            n = min(chunk, frame_begin - f)
            an = self.analyze(track, f, n, want_logspec=False)
            sc = self.scan(an, params, state)
            state = tuple(float(v) for v in sc.state.cpu().numpy())
            f += n
        return state

    # ------------------------------------------------------------------ raster
    def col2samp(self, W: int):
        if W not in self._tables:
            self._tables[W] = self.torch.from_numpy(waveform_col2samp(W)).to(self.device)
        return self._tables[W]

    def spans(self, kind: str, an: Analysis, sc: Scan, W: int, H: int, first: int = 0, count: int | None = None,
              out=None):
        """Column spans uint32 [count, K, W] for frames first .. first+count of the analysed range."""
        torch = self.torch
        if count is None:
            count = an.n_frames - first
        K = {"spectrum_linear": _lib.K_SPECTRUM, "particles": _lib.K_PARTICLES,
             "waveform_linear": _lib.K_WAVEFORM}.get(kind)
        if K is None:
            raise NotImplementedError(f"mode kind {kind!r} is not implemented on the HIP path")
        if out is None:
            out = self.empty((count, K, W), torch.int32)
        h, st = self.ctx.handle, self.stream()
        energy = sc.energy[first:first + count]
        if kind == "spectrum_linear":
            rc = self.lib.aur_spans_spectrum_linear(h, _ptr(an.logspec[first:first + count]), _ptr(energy), count,
                                                    W, H, _ptr(out), st)
        elif kind == "particles":
            rc = self.lib.aur_spans_particles(h, _ptr(an.logspec[first:first + count]), count, W, H, _ptr(out), st)
        else:
            if an.samples is None:
                raise ValueError("waveform mode needs analyze(..., want_samples=True)")
            rc = self.lib.aur_spans_waveform_linear(h, _ptr(an.samples[first:first + count]), _ptr(self.col2samp(W)),
                                                    _ptr(energy), count, W, H, _ptr(out), st)
        _lib.check(rc, f"aur_spans_{kind}")
        return out, K

    def raster_work(self, count: int, W: int):
        """Work area of the linear fill for `count` frames of width W (grown on demand, reused across calls on the
        engine's stream: launches on one stream are ordered, so the next batch may overwrite it)."""
        need = C.c_size_t(0)
        _lib.check(self.lib.aur_raster_work_bytes(int(count), int(W), C.byref(need)), "aur_raster_work_bytes")
        if self._work is None or self._work.numel() < need.value:
            self._work = self.empty((max(need.value, 16),), self.torch.uint8)
        return self._work

    def summarize(self, spans, K: int, W: int, H: int, work=None):
        """Column summaries of `spans` for the linear fill (aur_span_summary) into the work area; returns it."""
        count = int(spans.shape[0])
        if work is None:
            work = self.raster_work(count, W)
        if W % 16 == 0:
            _lib.check(self.lib.aur_span_summary(self.ctx.handle, _ptr(spans), K, count, W, H, _ptr(work),
                                                 work.numel(), self.stream()), "aur_span_summary")
        return work

    def fill(self, spans, K: int, pen_rgb, W: int, H: int, out=None, work=None, work_ready=False):
        torch = self.torch
        count = int(spans.shape[0])
        if out is None:
            out = self.empty((count, H, W, 3), torch.uint8)
        if work is None:
            work = self.raster_work(count, W)
            work_ready = False
        _lib.check(self.lib.aur_raster_spans_ws(self.ctx.handle, _ptr(spans), K, _ptr(pen_rgb), count, W, H,
                                                _ptr(out), _ptr(work), work.numel(), int(bool(work_ready)),
                                                self.stream()), "aur_raster_spans_ws")
        return out

    def fill_tiles(self, spans, K: int, pen_rgb, W: int, H: int, out=None):
        """The row-tile fill (no work area): aur_raster_spans."""
        count = int(spans.shape[0])
        if out is None:
            out = self.empty((count, H, W, 3), self.torch.uint8)
        _lib.check(self.lib.aur_raster_spans(self.ctx.handle, _ptr(spans), K, _ptr(pen_rgb), count, W, H,
                                             _ptr(out), self.stream()), "aur_raster_spans")
        return out

    def render(self, mode: str, an: Analysis, sc: Scan, W: int, H: int, first: int = 0, count: int | None = None,
               out=None, spans_out=None):
        """Frames uint8 [count, H, W, 3] (rgb24, tightly packed) for part of an analysed range."""
        kind = mode_kind(mode)
        if count is None:
            count = an.n_frames - first
        spans, K = self.spans(kind, an, sc, W, H, first, count, out=spans_out)
        return self.fill(spans, K, sc.pen_rgb[first:first + count], W, H, out=out)

    # ------------------------------------------------------------------ Spectrum - Radial
    def radial_prepare(self, an: Analysis, cfg: RadialConfig, state=None) -> RadialPrep:
        """Ring values of every analysed frame (stage 1 + temporal EMA).  state: None at track start, else the
        float64 [L1] EMA state carried from the frames before ``an.frame_begin``."""
        torch = self.torch
        F, L1 = an.n_frames, cfg.L1
        vals1 = self.empty((F, L1), torch.float64)
        h, st = self.ctx.handle, self.stream()
        _lib.check(self.lib.aur_radial_stage1(h, _ptr(an.logspec), F, int(cfg.radial_mirror),
                                              int(cfg.radial_wave_smoothness), _ptr(vals1), L1, st), "aur_radial_stage1")
        st_t = self.empty((L1,), torch.float64)
        if cfg.radial_temporal_alpha > 0.0:
            has = state is not None
            if has:
                st_t.copy_(state)
            _lib.check(self.lib.aur_radial_ema(h, _ptr(vals1), F, L1, float(cfg.radial_temporal_alpha), int(has),
                                               _ptr(st_t), st), "aur_radial_ema")
        key = ("radial", cfg.L2, cfg.radial_rotation_deg)
        if key not in self._tables:
            self._tables[key] = (torch.from_numpy(angle_table(cfg.L2, cfg.radial_rotation_deg)).to(self.device),
                                   torch.from_numpy(angle_table(360, cfg.radial_rotation_deg)).to(self.device))
        cs, cs360 = self._tables[key]
        return RadialPrep(cfg, vals1, st_t, cs, cs360, {}, self._center_image(cfg.center_image_path))

    def _center_image(self, path):
        """Device copy of the centre image, decoded once per engine; None when there is none or it cannot be read
        (`set_center_image` returns False and the disc keeps its plain fill, rt_widget.py:226-242)."""
        if not path:
            return None
        try:                                       # a file rewritten under the same name is another image
            st = os.stat(path)
            stamp = (st.st_mtime_ns, st.st_size)
        except OSError:
            stamp = None
        key = ("center_image", str(path), stamp)
        if key not in self._tables:
            from .images import load_image_rgba, premultiply_argb
            try:
                self._tables[key] = self.upload_image(premultiply_argb(load_image_rgba(path)))
            except Exception as e:                                   # noqa: BLE001 - any decode failure: no image
                import logging
                logging.getLogger(__name__).warning("centre image %r not loaded: %s", path, e)
                self._tables[key] = None
        return self._tables[key]

    # ------------------------------------------------------------------ background image
    def compose_background(self, bg, W: int, H: int):
        """The frame every export frame starts from when a background image is set: uint8 [H, W, 3] on the device
        (`paint_frame` rt_widget.py:829-830 -> `_draw_background` :505-545), or None when the file cannot be read
        (`QImage(path).isNull()`: the reference then draws no background, :508-509)."""
        from .images import load_image_rgba, premultiply_argb
        try:
            img = self.upload_image(premultiply_argb(load_image_rgba(bg.path)))
        except Exception as e:                                       # noqa: BLE001
            import logging
            logging.getLogger(__name__).warning("background image %r not loaded: %s", bg.path, e)
            return None
        ih, iw = int(img.shape[0]), int(img.shape[1])
        sw, sh, tile, x, y, dim_alpha = bg.plan(iw, ih, W, H)
        scaled = img if tile else self.scale_image(img, sw, sh)
        base = self.empty((H, W, 3), self.torch.uint8)
        _lib.check(self.lib.aur_compose_background(self.ctx.handle, _ptr(scaled), sw, sh, int(tile), int(x), int(y),
                                                   int(dim_alpha), W, H, _ptr(base), self.stream()),
                   "aur_compose_background")
        return base

    def set_background(self, base, W: int = 0, H: int = 0):
        """Frames rendered from now on start from ``base`` uint8 [H, W, 3] (None: the plain (13, 13, 18) fill)."""
        _lib.check(self.lib.aur_set_background(self.ctx.handle, _ptr(base), int(W), int(H)), "aur_set_background")
        self._base = base

    def carried_radial_state(self, track: Track, frame_begin: int, cfg: RadialConfig, chunk: int = 16384):
        """EMA state of the radial ring in front of ``frame_begin`` (analysis-only pre-pass over [0, frame_begin))."""
        if frame_begin <= 0 or cfg.radial_temporal_alpha <= 0.0:
            return None
        state, f = None, 0
        while f < frame_begin:
            n = min(chunk, frame_begin - f)
            prep = self.radial_prepare(self.analyze(track, f, n), cfg, state)
            state = prep.state
            f += n
        return state

    def radial_path_vertices(self, prep: RadialPrep, sc: Scan, W: int, H: int, first: int, count: int):
        """Radial path float64 [count, L2 + 1, 2] of frames first .. first + count."""
        cfg = prep.cfg
        verts = self.empty((count, cfg.L2 + 1, 2), self.torch.float64)
        _lib.check(self.lib.aur_radial_vertices(
            self.ctx.handle, _ptr(prep.vals1[first:first + count]), _ptr(sc.energy[first:first + count]), count, cfg.L1,
            int(cfg.radial_smooth_amount), _ptr(prep.cos_sin), W, H, _ptr(verts), cfg.L2, self.stream()),
            "aur_radial_vertices")
        return verts

    def radial_vertices(self, prep: RadialPrep, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int):
        """(path float64 [count, L2 + 1, 2], envelope float64 [count, 361, 2]) of frames first .. first + count."""
        return (self.radial_path_vertices(prep, sc, W, H, first, count),
                self._envelope(prep, an, sc, W, H, first, count))

    @staticmethod
    def _envelope_params(cfg: RadialConfig):
        return _lib.EnvelopeParams(int(cfg.radial_mirror), int(cfg.edge_waviness), int(cfg.feather_audio_enabled),
                                   int(cfg.feather_audio_amount), float(cfg.feather_sensitivity),
                                   float(cfg.waveform_sensitivity), float(np.deg2rad(cfg.radial_rotation_deg)))

    def feather_ring_vertices(self, prep: RadialPrep, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int):
        """Feather ring paths float64 [count, 6, 361, 2] (rt_widget.py:1098-1114)."""
        ep = self._envelope_params(prep.cfg)
        rings = self.empty((count, 6, 361, 2), self.torch.float64)
        _lib.check(self.lib.aur_feather_ring_vertices(
            self.ctx.handle, _ptr(an.logspec[first:first + count]), _ptr(sc.energy[first:first + count]),
            _ptr(sc.phase[first:first + count]), count, C.byref(ep), _ptr(prep.cos_sin360), W, H, _ptr(rings),
            self.stream()), "aur_feather_ring_vertices")
        return rings

    def apply_fill(self, fill: FillConfig, verts, sc: Scan, first: int, count: int, stroke, disc, disc_static, frames):
        """`_apply_fill`: the polygon of the closed path ``verts`` in the fill's colour wherever neither the stroke nor the
        disc paints afterwards, in the frames whose energy reaches the threshold (in place on ``frames``)."""
        torch = self.torch
        H, W = int(frames.shape[1]), int(frames.shape[2])
        W32 = (W + 31) // 32
        h, st = self.ctx.handle, self.stream()
        plane = self.empty((count, 1, H, W32), torch.int32)
        _lib.check(self.lib.aur_mask_polygon(h, _ptr(verts), int(verts.shape[1]), count, W, H, _ptr(plane), st),
                   "aur_mask_polygon")
        enable = (sc.energy[first:first + count] >= float(fill.threshold)).to(torch.uint8)     # rt_widget.py:857
        _lib.check(self.lib.aur_plane_exclude(h, _ptr(plane), _ptr(stroke), _ptr(disc), int(bool(disc_static)),
                                              _ptr(enable), count, W, H, st), "aur_plane_exclude")
        if self._base is None:
            # painted straight onto the plain background: the result is one colour, composed on the host
            return self.blend_planes(plane, bytes(fill.colour_over_background()) + b"\xff", frames)
        # over a background image the blend is per pixel
        if fill.blend == "add":
            pm = torch.tensor([_div255(int(c) * int(fill.rgba[3])) for c in fill.rgba[:3]] + [0], dtype=torch.uint8,
                              device=self.device).repeat(count, 1).reshape(count, 1, 4).contiguous()
            _lib.check(self.lib.aur_add_planes(h, _ptr(plane), 1, _ptr(pm), None, None, 1, count, W, H, _ptr(frames), st),
                       "aur_add_planes")
            return frames
        if fill.blend == "multiply":
            _lib.check(self.lib.aur_multiply_plane(h, _ptr(plane), bytes(int(v) for v in fill.rgba), count, W, H,
                                                   _ptr(frames), st), "aur_multiply_plane")
            return frames
        return self.blend_planes(plane, bytes(int(v) for v in fill.rgba), frames)

    def apply_shadow(self, shadow: ShadowConfig, verts, closed: bool, excl_a, excl_b, excl_b_static, frames,
                     round_join: bool = False, half_width=None, offset=None):
        """`_draw_shadow_path` of ``verts`` float64 [count, n, 2]: translated wide stroke in QColor(0, 0, 0, alpha),
        source-over on the frame wherever the planes ``excl_a`` / ``excl_b`` (what the reference paints afterwards
        with opaque colours) have no bit (in place)."""
        torch = self.torch
        count, H, W = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        W32 = (W + 31) // 32
        h, st = self.ctx.handle, self.stream()
        plane = self.empty((count, 1, H, W32), torch.int32)
        dx, dy = shadow.offset if offset is None else offset
        _lib.check(self.lib.aur_mask_stroke_path(h, _ptr(verts), int(verts.shape[1]), count, W, H,
                                                 float(half_width if half_width is not None else shadow.half_width),
                                                 float(dx), float(dy), int(bool(closed)), int(bool(round_join)),
                                                 _ptr(plane), st), "aur_mask_stroke_path")
        _lib.check(self.lib.aur_plane_exclude(h, _ptr(plane), _ptr(excl_a), _ptr(excl_b), int(bool(excl_b_static)),
                                              None, count, W, H, st), "aur_plane_exclude")
        return self.blend_planes(plane, bytes((0, 0, 0, shadow.alpha)), frames)

    def apply_glow(self, glow: GlowConfig, verts, sc: Scan, first: int, count: int, excl_a, excl_b, excl_b_static,
                   frames):
        """`_draw_glow_offscreen` of the closed path ``verts`` float64 [count, n, 2]: every layer's antialiased RoundJoin
        stroke from one distance pass, added (CompositionMode_Plus) in the layer's premultiplied colour wherever the
        planes ``excl_a`` / ``excl_b`` (stroke and disc, painted afterwards with opaque colours) have no bit (in place)."""
        torch = self.torch
        H, W = int(frames.shape[1]), int(frames.shape[2])
        W32 = (W + 31) // 32
        h, st = self.ctx.handle, self.stream()
        table = glow.layer_table()
        K = len(table)
        # layer colours of the whole scan, once (12 B per frame and layer from the host-side expression)
        key = (tuple(glow.rgba), glow.radius, glow.strength, glow.layers)
        cache = getattr(sc, "_glow_pm", None)
        if cache is None or cache[0] != key:
            cache = (key, torch.from_numpy(glow.layer_premultiplied(sc.energy.cpu().numpy())).to(self.device))
            sc._glow_pm = cache
        pm = cache[1][first:first + count]
        nv = int(verts.shape[1])
        need = C.c_size_t(0)
        _lib.check(self.lib.aur_glow_work_bytes(count, nv, C.byref(need)), "aur_glow_work_bytes")
        if self._glow_work is None or self._glow_work.numel() < need.value:
            self._glow_work = self.empty((need.value,), torch.uint8)
        hw = (C.c_double * K)(*[width / 2.0 for width, _ in table])
        _lib.check(self.lib.aur_glow_path(h, _ptr(verts), nv, count, W, H, K, hw, _ptr(pm), _ptr(excl_a), _ptr(excl_b),
                                          int(bool(excl_b_static)), _ptr(self._glow_work), self._glow_work.numel(),
                                          _ptr(frames), st), "aur_glow_path")
        return frames

    # QColor(13, 13, 18) with alpha max(10, 70 - 10 k), k = 1..6 (rt_widget.py:1099, :1116)
    FEATHER_RGBA = bytes(v for k in range(1, 7) for v in (13, 13, 18, max(10, 70 - 10 * k)))
    DISC_RGBA = bytes((30, 30, 40, 255))                                    # rt_widget.py:1092/1094

    def blend_planes(self, planes, rgba: bytes, frames):
        """Source-over of constant colours, plane k of ``planes`` int32 [n, K, H, W32] in plane order, in place."""
        n, K, H, W32 = (int(v) for v in planes.shape)
        W = int(frames.shape[2])
        assert len(rgba) == 4 * K and frames.shape[0] == n and frames.shape[1] == H and (W + 31) // 32 == W32
        _lib.check(self.lib.aur_blend_planes(self.ctx.handle, _ptr(planes), K, rgba, n, W, H, _ptr(frames),
                                             self.stream()), "aur_blend_planes")
        return frames

    def feather_blend(self, prep: RadialPrep, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int, frames,
                      with_disc: bool = False):
        """Blends the six translucent rings of frames first .. first + count into ``frames`` (in place).  with_disc:
        the centre disc (30, 30, 40) is composited first from its own plane (frames of a mode that has no disc)."""
        h, st = self.ctx.handle, self.stream()
        K = 7 if with_disc else 6
        step = 65535 // K
        # planes of the box the disc and its rings stay in, not of the frame (a quarter of a 1080p frame)
        bx, by, bw, bh = self.feather_box(prep.cfg, W, H)
        rgba = (self.DISC_RGBA if with_disc else b"") + self.FEATHER_RGBA
        for off in range(0, count, step):
            n = min(step, count - off)
            rings = self.feather_ring_vertices(prep, an, sc, W, H, first + off, n)
            if with_disc:
                env = self._envelope(prep, an, sc, W, H, first + off, n)
                rings = self.torch.cat([env[:, None], rings], dim=1).contiguous()
            planes = self.empty((n, K, bh, (bw + 31) // 32), self.torch.int32)
            _lib.check(self.lib.aur_mask_polygon_box(h, _ptr(rings), 361, K * n, W, H, bx, by, bw, bh, _ptr(planes), st),
                       "aur_mask_polygon_box")
            _lib.check(self.lib.aur_blend_planes_box(h, _ptr(planes), K, rgba, n, W, H, bx, by, bw, bh,
                                                     _ptr(frames[off:off + n]), st), "aur_blend_planes_box")
        return frames

    @staticmethod
    def feather_box(cfg: RadialConfig, W: int, H: int):
        """Pixel box (x, y, w, h; x a multiple of 32) that holds the envelope path and the six feather rings whatever the
        audio does.  With m = min(w, h): the raw radius is at most inner + a, inner = 0.22 m, a = (0.025 waviness + 0.06
        audio amount) m (|fbm|, |sin|, spec_norm and the energy factor are all <= 1; rt_widget.py:1021-1038); a ring
        scales it by at most 1.09 (:1103); the envelope's soft limit maps g0 = a / (r_max - inner) to tanh(1.25 g0) /
        tanh(1.25) <= min(1 / tanh(1.25), 1.25 g0 / tanh(1.25)) (:1040-1044)."""
        m = float(min(W, H))
        noise = cfg.edge_waviness / 100.0
        audio = cfg.feather_audio_amount / 100.0 if cfg.feather_audio_enabled else 0.0
        a = (0.025 * noise + 0.06 * audio) * m
        ring = (0.22 * m + a) * 1.09
        t = float(np.tanh(1.25))
        env = 0.22 * m + min(0.26 * m / t, 1.25 * a / t)
        r = max(ring, env) + 3.0
        cx, cy = W / 2.0, H / 2.0
        x0 = max(0, int(np.floor(cx - r))) & ~31
        y0 = max(0, int(np.floor(cy - r)))
        x1, y1 = min(W, int(np.ceil(cx + r)) + 1), min(H, int(np.ceil(cy + r)) + 1)
        return x0, y0, x1 - x0, y1 - y0

    def _envelope(self, prep: RadialPrep, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int):
        cfg = prep.cfg
        ep = self._envelope_params(cfg)
        env = self.empty((count, 361, 2), self.torch.float64)
        _lib.check(self.lib.aur_envelope_vertices(
            self.ctx.handle, _ptr(an.logspec[first:first + count]), _ptr(sc.energy[first:first + count]),
            _ptr(sc.phase[first:first + count]), count, C.byref(ep), _ptr(prep.cos_sin360), W, H, _ptr(env),
            self.stream()), "aur_envelope_vertices")
        return env

    # ---- centre disc: antialiased clip + fill / centre image (rt_widget.py:1056-1096) ----------------------------------
    DISC_ARGB = (255 << 24) | (30 << 16) | (30 << 8) | 40                   # QColor(30, 30, 40), rt_widget.py:1092/1094

    @staticmethod
    def disc_box(W: int, H: int, static: bool):
        """Pixel box the envelope path lies in.  r = inner + g (r_max - inner) with g = tanh(1.25 g0) / tanh(1.25) <=
        1 / tanh(1.25) (rt_widget.py:1040-1044), so r <= 0.22 m + 0.26 m / tanh(1.25) < 0.5266 m, m = min(w, h); a
        frame-independent disc has r = inner = 0.22 m."""
        m = min(W, H)
        r = (0.22 * m if static else 0.5266 * m) + 3.0
        cx, cy = W / 2.0, H / 2.0
        x0, y0 = max(0, int(np.floor(cx - r))), max(0, int(np.floor(cy - r)))
        x1, y1 = min(W, int(np.ceil(cx + r)) + 1), min(H, int(np.ceil(cy + r)) + 1)
        return x0, y0, x1 - x0, y1 - y0

    def upload_image(self, argb_pm: np.ndarray):
        """Premultiplied 0xAARRGGBB uint32 [h, w] -> device tensor (int32 view of the same bits)."""
        a = np.ascontiguousarray(argb_pm, dtype=np.uint32)
        return self.torch.from_numpy(a.view(np.int32)).to(self.device)

    def scale_image(self, img, dw: int, dh: int):
        """QImage.scaled(dw, dh, IgnoreAspectRatio, SmoothTransformation) of a device image int32 [h, w]."""
        sh, sw = int(img.shape[0]), int(img.shape[1])
        if (dw, dh) == (sw, sh):
            return img                                                    # QImage::scaled returns *this
        out = self.empty((dh, dw), self.torch.int32)
        _lib.check(self.lib.aur_image_scale(self.ctx.handle, _ptr(img), sw, sh, _ptr(out), int(dw), int(dh),
                                            self.stream()), "aur_image_scale")
        return out

    def disc_composite(self, env, env_static: bool, count: int, W: int, H: int, box, frames=None, source=None,
                       skip_full=False, full_plane=None):
        """aur_disc_composite on ``frames`` uint8 [count, H, W, 3] (in place) and / or into ``full_plane``."""
        src = source if source is not None else _lib.DiscSource(0, self.DISC_ARGB, None, 0, 0, None, 0)
        _lib.check(self.lib.aur_disc_composite(
            self.ctx.handle, _ptr(env), int(env.shape[1]), int(bool(env_static)), int(count), W, H, *box, C.byref(src),
            int(bool(skip_full)), _ptr(full_plane), _ptr(frames), self.stream()), "aur_disc_composite")
        return frames

    def _static_disc(self, prep: RadialPrep, an: Analysis, sc: Scan, W: int, H: int, first: int):
        """Everything a frame-independent disc needs, once per frame size: its envelope, box, the bit plane of its fully
        covered pixels (painted by the plane fill; what later translucent layers must leave alone) and the list of its
        partially covered pixels with their effective coverage (fill x clip)."""
        torch = self.torch
        key = (W, H)
        if key not in prep.static_disc:
            W32 = (W + 31) // 32
            env = self._envelope(prep, an, sc, W, H, first, 1)
            box = self.disc_box(W, H, True)
            full = torch.zeros((1, H, W32), dtype=torch.int32, device=self.device)
            # coverage read-out: white at the disc's effective coverage on a black frame leaves exactly that coverage
            probe = torch.zeros((1, H, W, 3), dtype=torch.uint8, device=self.device)
            white = _lib.DiscSource(0, 0xFFFFFFFF, None, 0, 0, None, 0)
            self.disc_composite(env, True, 1, W, H, box, frames=probe, source=white, full_plane=full)
            ce = probe[0, :, :, 0].reshape(-1)
            idx = torch.nonzero((ce > 0) & (ce < 255)).reshape(-1)
            entry = dict(env=env, box=box, full=full, rows=self.disc_rows(full, W, H),
                         edge_index=idx.to(torch.int32).contiguous(),
                         edge_cov=ce[idx].contiguous(), placement=None, scaled=None)
            if prep.image is not None:
                ih, iw = int(prep.image.shape[0]), int(prep.image.shape[1])
                if prep.cfg.center_motion == 0:
                    # the placement does not depend on the frame either: scale the image once
                    pl = self.image_placement(prep, env, True, sc, first, 1)
                    ow, oh = (int(v) for v in pl[0, :2].cpu().numpy())
                    entry["placement"] = pl
                    entry["scaled"] = self.scale_image(prep.image, ow, oh)
            prep.static_disc[key] = entry
        return prep.static_disc[key]

    def image_placement(self, prep: RadialPrep, env, env_static: bool, sc: Scan, first: int, count: int):
        """int32 [count, 4] = out_w, out_h, dx, dy of the centre image (rt_widget.py:1065-1088)."""
        cfg = prep.cfg
        ih, iw = int(prep.image.shape[0]), int(prep.image.shape[1])
        out = self.empty((count, 4), self.torch.int32)
        _lib.check(self.lib.aur_center_image_placement(
            self.ctx.handle, _ptr(env), int(env.shape[1]), int(bool(env_static)), _ptr(sc.energy[first:first + count]),
            _ptr(sc.phase[first:first + count]), count, iw, ih, int(cfg.center_image_zoom), int(cfg.center_motion),
            _ptr(out), self.stream()), "aur_center_image_placement")
        return out

    def draw_disc(self, prep: RadialPrep, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int, frames):
        """The centre disc of frames first .. first + count on ``frames`` (in place), after everything the reference
        paints before it (rt_widget.py:1056-1096)."""
        cfg = prep.cfg
        if cfg.disc_static:
            sd = self._static_disc(prep, an, sc, W, H, first)
            if prep.image is None:
                # interior painted by the plane fill from sd["full"]; the rim is a short list of blended pixels
                n = int(sd["edge_index"].numel())
                _lib.check(self.lib.aur_blend_pixels(self.ctx.handle, _ptr(sd["edge_index"]), _ptr(sd["edge_cov"]), n,
                                                     self.DISC_ARGB, count, W, H, _ptr(frames), self.stream()),
                           "aur_blend_pixels")
                return frames
            if sd["scaled"] is not None:
                sc_img = sd["scaled"]
                src = _lib.DiscSource(2, 0, sc_img.data_ptr(), int(sc_img.shape[1]), int(sc_img.shape[0]),
                                      sd["placement"].data_ptr(), 1)
                return self.disc_composite(sd["env"], True, count, W, H, sd["box"], frames=frames, source=src)
            env, env_static, box = sd["env"], True, sd["box"]
        else:
            env, env_static, box = self._envelope(prep, an, sc, W, H, first, count), False, self.disc_box(W, H, False)
        if prep.image is None:
            return self.disc_composite(env, env_static, count, W, H, box, frames=frames)
        pl = self.image_placement(prep, env, env_static, sc, first, count)
        src = _lib.DiscSource(1, 0, prep.image.data_ptr(), int(prep.image.shape[1]), int(prep.image.shape[0]),
                              pl.data_ptr(), 0)
        return self.disc_composite(env, env_static, count, W, H, box, frames=frames, source=src)

    def radial_planes(self, prep: RadialPrep, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int,
                      planes_out=None):
        """(stroke planes int32 [count, H, W32], disc plane int32 [1, H, W32], True).

        The disc plane holds what the plane fill may paint in (30, 30, 40) ahead of the antialiased composite and what
        translucent layers must leave alone: the fully covered pixels of a frame-independent disc that is filled with
        the plain colour (export defaults: edge_waviness 0, audio term off, no centre image); nothing otherwise -- the
        composite (draw_disc) then paints the whole disc."""
        torch = self.torch
        W32 = (W + 31) // 32
        h, st = self.ctx.handle, self.stream()
        if planes_out is None:
            planes_out = self.empty((2, count, H, W32), torch.int32)
        verts = self.radial_path_vertices(prep, sc, W, H, first, count)
        _lib.check(self.lib.aur_mask_stroke_closed(h, _ptr(verts), int(verts.shape[1]), count, W, H,
                                                   _ptr(planes_out[0]), st), "aur_mask_stroke_closed")
        if prep.cfg.disc_static and prep.image is None:
            return planes_out[0], self._static_disc(prep, an, sc, W, H, first)["full"], True
        return planes_out[0], self._no_disc_plane(W, H), True

    def _no_disc_plane(self, W: int, H: int):
        key = ("nodisc", W, H)
        if key not in self._tables:
            self._tables[key] = self.torch.zeros((1, H, (W + 31) // 32), dtype=self.torch.int32, device=self.device)
        return self._tables[key]

    def _no_disc_rows(self, W: int, H: int):
        """Row table of the empty disc plane: every row empty."""
        key = ("nodisc_rows", H)
        if key not in self._tables:
            self._tables[key] = self.torch.zeros((1, H), dtype=self.torch.int32, device=self.device)
        return self._tables[key]

    def fill_planes(self, planes, pen_rgb, W: int, H: int, out=None):
        stroke, disc, static = planes
        count = int(stroke.shape[0])
        if out is None:
            out = self.empty((count, H, W, 3), self.torch.uint8)
        ev = self.fill_events
        if ev is not None:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(self.lib.aur_raster_planes(self.ctx.handle, _ptr(stroke), _ptr(disc), int(static), _ptr(pen_rgb),
                                              count, W, H, _ptr(out), self.stream()), "aur_raster_planes")
        if ev is not None:
            e1.record()
            ev.append((e0, e1, count))
        return out

    def disc_rows(self, disc, W: int, H: int):
        """Row table uint32 [n, H] of disc planes int32 [n, H, W32] (aur_disc_rows)."""
        rows = self.empty((int(disc.shape[0]), H), self.torch.int32)
        _lib.check(self.lib.aur_disc_rows(self.ctx.handle, _ptr(disc), int(disc.shape[0]), W, H, _ptr(rows),
                                          self.stream()), "aur_disc_rows")
        return rows

    def raster_closed_path(self, verts, disc, static, pen_rgb, W: int, H: int, out, rows=None):
        """Closed-path frames without a stroke plane (aur_raster_closed_path): background / disc rows, then the stroke
        painted straight into the frames.  ``rows``: the disc's row table when the caller keeps one (a disc that does
        not change from batch to batch), else it is built inside the call."""
        count = int(verts.shape[0])
        ev = self.fill_events
        if ev is not None:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(self.lib.aur_raster_closed_path(self.ctx.handle, _ptr(verts), int(verts.shape[1]), count, W, H,
                                                   _ptr(disc), int(static), _ptr(rows), _ptr(pen_rgb), _ptr(out),
                                                   self.stream()), "aur_raster_closed_path")
        if ev is not None:
            e1.record()
            ev.append((e0, e1, count))
        return out

    def _direct_stroke_ok(self, W: int, H: int, count: int, out, decorated: bool):
        """No translucent layer needs the stroke plane as an exclusion mask, and frame shape and batch are ones
        aur_raster_closed_path takes (its own preconditions, csrc/radial.hip: the linear fill's shapes, the row-tile
        A/B knob AUR_FILL_MODE=1 off, 32-bit piece indices); everything else goes through the stroke plane."""
        upf = (W * 3 // 16) * H                      # 16-byte pieces per frame
        return (self.direct_stroke and not decorated and W % 32 == 0 and out is not None and out.is_contiguous()
                and out.data_ptr() % 16 == 0 and self.fill_mode != 1
                and 256 <= upf < (1 << 26) and count * upf + 256 < (1 << 31))

    def render_radial(self, prep: RadialPrep, an: Analysis, sc: Scan, W: int, H: int, first: int = 0,
                      count: int | None = None, out=None, planes_out=None, fill: FillConfig | None = None,
                      shadow: ShadowConfig | None = None, glow: GlowConfig | None = None):
        """Frames uint8 [count, H, W, 3] of "Spectrum - Radial" (paint_frame, rt_widget.py:821-852).  The reference's
        order is fill, path shadow, glow, stroke, envelope shadow, disc, feather rings; stroke and disc are opaque, so
        the translucent layers are applied afterwards where those two do not cover them."""
        if count is None:
            count = an.n_frames - first
        want_fill = fill is not None and fill.enabled
        want_shadow = shadow is not None and shadow.active
        want_glow = glow is not None and glow.active
        if out is None:
            out = self.empty((count, H, W, 3), self.torch.uint8)
        if self._direct_stroke_ok(W, H, count, out, want_fill or want_shadow or want_glow):
            if prep.cfg.disc_static and prep.image is None:
                sd = self._static_disc(prep, an, sc, W, H, first)
                disc, rows = sd["full"], sd["rows"]          # the disc does not change: its row table is built once
            else:
                disc, rows = self._no_disc_plane(W, H), self._no_disc_rows(W, H)
            self.raster_closed_path(self.radial_path_vertices(prep, sc, W, H, first, count), disc, True,
                                    sc.pen_rgb[first:first + count], W, H, out, rows=rows)
            planes = None
        else:
            planes = self.radial_planes(prep, an, sc, W, H, first, count, planes_out)
            self.fill_planes(planes, sc.pen_rgb[first:first + count], W, H, out)
        if want_fill or want_shadow or want_glow:
            verts = self.radial_path_vertices(prep, sc, W, H, first, count)
        if want_fill:
            self.apply_fill(fill, verts, sc, first, count, planes[0], planes[1], planes[2], out)
        if want_shadow:
            self.apply_shadow(shadow, verts, True, planes[0], planes[1], planes[2], out)          # :978-979
        if want_glow:
            self.apply_glow(glow, verts, sc, first, count, planes[0], planes[1], planes[2], out)  # :980
        if want_shadow:
            env = self._envelope(prep, an, sc, W, H, first, count)
            self.apply_shadow(shadow, env, False, None, planes[1], planes[2], out)                # :1053-1054
        self.draw_disc(prep, an, sc, W, H, first, count, out)                                     # :1056-1096
        if prep.cfg.feather_enabled:
            self.feather_blend(prep, an, sc, W, H, first, count, out)
        return out

    # ------------------------------------------------------------------ Waveform - Linear with its shadow
    def wave_linear_vertices(self, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int):
        """Polyline float64 [count, W, 2] of "Waveform - Linear" (rt_widget.py:904-913)."""
        if an.samples is None:
            raise ValueError("Waveform - Linear needs analyze(..., want_samples=True)")
        verts = self.empty((count, W, 2), self.torch.float64)
        _lib.check(self.lib.aur_wave_linear_vertices(
            self.ctx.handle, _ptr(an.samples[first:first + count]), _ptr(self.col2samp(W)),
            _ptr(sc.energy[first:first + count]), count, W, H, _ptr(verts), self.stream()), "aur_wave_linear_vertices")
        return verts

    def render_wave_linear_shadow(self, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int,
                                  shadow: ShadowConfig, out=None):
        """"Waveform - Linear" with its shadow (rt_widget.py:915-932): the shadow is the polyline translated by
        (distance, distance), stroked with a RoundCap / RoundJoin pen of width max(2, min(10, blur)) in
        QColor(0, 0, 0, alpha); the polyline itself is stroked through a bit plane here (same coverage rule as the
        column spans of the plain mode: pixel centre inside the width-2 SquareCap / BevelJoin outline)."""
        torch = self.torch
        W32 = (W + 31) // 32
        verts = self.wave_linear_vertices(an, sc, W, H, first, count)
        stroke = self.empty((count, H, W32), torch.int32)
        _lib.check(self.lib.aur_mask_stroke_path(self.ctx.handle, _ptr(verts), W, count, W, H, 1.0, 0.0, 0.0, 0, 0,
                                                 _ptr(stroke), self.stream()), "aur_mask_stroke_path")
        key = ("nodisc", W, H)
        if key not in self._tables:
            self._tables[key] = torch.zeros((1, H, W32), dtype=torch.int32, device=self.device)
        out = self.fill_planes((stroke, self._tables[key], True), sc.pen_rgb[first:first + count], W, H, out)
        self.apply_shadow(shadow, verts, False, stroke, None, True, out, round_join=True,
                          half_width=shadow.linear_pen_width / 2.0, offset=shadow.linear_offset)
        return out

    # ------------------------------------------------------------------ Waveform - Circular
    def wave_circular_vertices(self, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int,
                               rotation_deg: float = 0.0):
        """Closed path float64 [count, 1024, 2] of frames first .. first + count (rt_widget.py:871-896)."""
        if an.samples is None:
            raise ValueError("Waveform - Circular needs analyze(..., want_samples=True)")
        key = ("wavecirc", float(rotation_deg))
        if key not in self._tables:
            self._tables[key] = self.torch.from_numpy(wave_circular_angle_table(rotation_deg)).to(self.device)
        verts = self.empty((count, BLOCK, 2), self.torch.float64)
        _lib.check(self.lib.aur_wave_circular_vertices(
            self.ctx.handle, _ptr(an.samples[first:first + count]), _ptr(sc.energy[first:first + count]), count,
            _ptr(self._tables[key]), W, H, _ptr(verts), self.stream()), "aur_wave_circular_vertices")
        return verts

    def wave_circular_planes(self, an: Analysis, sc: Scan, W: int, H: int, first: int, count: int,
                             rotation_deg: float = 0.0, planes_out=None):
        """(stroke planes int32 [count, H, W32], all-zero disc plane [1, H, W32], True) for fill_planes."""
        torch = self.torch
        W32 = (W + 31) // 32
        stroke = planes_out[0] if planes_out is not None else self.empty((count, H, W32), torch.int32)
        verts = self.wave_circular_vertices(an, sc, W, H, first, count, rotation_deg)
        _lib.check(self.lib.aur_mask_stroke_closed(self.ctx.handle, _ptr(verts), BLOCK, count, W, H, _ptr(stroke),
                                                   self.stream()), "aur_mask_stroke_closed")
        key = ("nodisc", W, H)
        if key not in self._tables:
            self._tables[key] = torch.zeros((1, H, W32), dtype=torch.int32, device=self.device)
        return stroke, self._tables[key], True

    def render_wave_circular(self, an: Analysis, sc: Scan, W: int, H: int, first: int = 0, count: int | None = None,
                             rotation_deg: float = 0.0, out=None, planes_out=None, fill: FillConfig | None = None,
                             shadow: ShadowConfig | None = None, glow: GlowConfig | None = None):
        """Frames uint8 [count, H, W, 3] of "Waveform - Circular" (paint_frame, rt_widget.py:821-852)."""
        if count is None:
            count = an.n_frames - first
        want_fill = fill is not None and fill.enabled
        want_shadow = shadow is not None and shadow.active
        want_glow = glow is not None and glow.active
        if out is None:
            out = self.empty((count, H, W, 3), self.torch.uint8)
        if self._direct_stroke_ok(W, H, count, out, want_fill or want_shadow or want_glow):
            self.raster_closed_path(self.wave_circular_vertices(an, sc, W, H, first, count, rotation_deg),
                                    self._no_disc_plane(W, H), True, sc.pen_rgb[first:first + count], W, H, out,
                                    rows=self._no_disc_rows(W, H))
            return out
        planes = self.wave_circular_planes(an, sc, W, H, first, count, rotation_deg, planes_out)
        self.fill_planes(planes, sc.pen_rgb[first:first + count], W, H, out)
        if want_fill or want_shadow or want_glow:
            verts = self.wave_circular_vertices(an, sc, W, H, first, count, rotation_deg)
        if want_fill:
            self.apply_fill(fill, verts, sc, first, count, planes[0], None, True, out)            # :898
        if want_shadow:
            self.apply_shadow(shadow, verts, True, planes[0], None, True, out)                    # :899-900
        if want_glow:
            self.apply_glow(glow, verts, sc, first, count, planes[0], None, True, out)            # :901
        return out

    # ------------------------------------------------------------------ streaming Analyzer step
    def stft_buffer(self, d_buffer, d_prev_mag, d_logspec, d_flux_sum):
        _lib.check(self.lib.aur_stft_buffer(self.ctx.handle, _ptr(d_buffer), _ptr(d_prev_mag), _ptr(d_logspec),
                                            _ptr(d_flux_sum), self.stream()), "aur_stft_buffer")


_engines = {}


def get_engine(device: int = 0) -> HipExportEngine:
    """Process-wide engine per device (contexts are cheap but tables upload once)."""
    if device not in _engines:
        _engines[device] = HipExportEngine(device)
    return _engines[device]
