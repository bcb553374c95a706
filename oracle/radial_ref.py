"""ORACLE (test infrastructure, not product code): CPU restatement of Aurora's "Spectrum - Radial" export frame.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Restates (paths relative to the reference root):

* ``smooth_closed``       <- ``widgets/rt_widget.py:1121-1137`` (``_smooth_closed``: upsample + wrapped box filter)
* ``RadialParams``        <- widget defaults ``rt_widget.py:59-67, :106-114`` as overridden by the exporter's
                             ``_apply_view_state`` (``export/exporter.py:153-209``): in an export ``radial_mirror``,
                             ``edge_waviness``, ``feather_audio_*`` default to off/0
* ``RadialTrack.path``    <- ``_draw_spectrum`` radial branch ``rt_widget.py:942-975`` (incl. the ``_radial_prev``
                             temporal EMA, which makes the path depend on every earlier frame)
* ``envelope_path``       <- ``_draw_center_image_and_mask`` ``rt_widget.py:1004-1051`` (361-vertex clip path)
* ``feather_ring_paths``  <- ``rt_widget.py:1098-1114``
* ``wave_circular_path``  <- ``_draw_waveform`` circular branch ``rt_widget.py:871-896`` ("Waveform - Circular")

Parity pin: ``tests/golden/drawlists_*.npz`` hold the ``drawPath`` and ``setClipPath`` vertices the reference
itself produced for "Spectrum - Radial" (``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` compares.
``tests/golden/wavecirc_*.npz`` hold the ``drawPath`` vertices of "Waveform - Circular" the same way.
Pixel coverage of these paths (``raster_ref.py``) is parity-unpinned against Qt, like the linear modes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

N_BARS = 80
N_ENV = 360


def smooth_closed(arr, amount):
    """rt_widget.py:1121-1137."""
    if arr is None or len(arr) == 0 or amount <= 0:
        return arr
    L = len(arr)
    up = int(1 + (amount // 25))
    a = np.asarray(arr, dtype=float)
    if up > 1:
        x = np.arange(L, dtype=float)
        xi = np.linspace(0.0, L - 1, L * up)
        a = np.interp(xi, x, a)
    half = int(amount * 0.06)
    win = 2 * half + 1
    if win > 1:
        pad = np.pad(a, (half, half), mode='wrap')
        ker = np.ones(win, dtype=float) / float(win)
        a = np.convolve(pad, ker, mode='valid')
    return a


@dataclass
class RadialParams:
    radial_rotation_deg: float = 0.0        # exporter.py:154
    radial_mirror: bool = False             # exporter.py:155 (export default False; widget default True)
    radial_smooth_amount: int = 50          # exporter.py:183 (radial_smooth is always True, rt_widget.py:196-198)
    radial_wave_smoothness: int = 50        # rt_widget.py:66
    radial_temporal_alpha: float = 0.3      # rt_widget.py:67
    edge_waviness: int = 0                  # exporter.py:161
    feather_enabled: bool = False           # exporter.py:160
    feather_audio_enabled: bool = False     # exporter.py:162
    feather_audio_amount: int = 0           # exporter.py:163
    feather_sensitivity: float = 2.0        # rt_widget.py:60
    waveform_sensitivity: float = 1.0       # rt_widget.py:59

    @classmethod
    def from_view_state(cls, vs, sensitivity=1.0):
        vs = vs or {}
        p = cls()
        p.radial_rotation_deg = float(vs.get('radial_rotation_deg', 0.0))
        p.radial_mirror = bool(vs.get('radial_mirror', False))
        p.radial_smooth_amount = max(0, min(100, int(vs.get('radial_smooth_amount', 50))))
        if 'radial_waveform_smoothness' in vs:
            p.radial_wave_smoothness = max(0, min(100, int(vs['radial_waveform_smoothness'])))
        if 'radial_temporal_smoothing' in vs:
            p.radial_temporal_alpha = max(0.0, min(0.95, int(vs['radial_temporal_smoothing']) / 100.0))
        p.edge_waviness = max(0, min(100, int(vs.get('edge_waviness', 0))))
        p.feather_enabled = bool(vs.get('feather_enabled', False))
        p.feather_audio_enabled = bool(vs.get('feather_audio_enabled', False))
        p.feather_audio_amount = max(0, min(100, int(vs.get('feather_audio_amount', 0))))
        p.waveform_sensitivity = max(0.05, float(sensitivity))
        return p


class RadialTrack:
    """Sequential state of the radial spectrum path (``_radial_prev``)."""

    def __init__(self, params: RadialParams | None = None):
        self.p = params or RadialParams()
        self.prev = None

    def values(self, spectrum):
        """spec_draw after mirror / spatial smoothing / temporal EMA / spatial smoothing (rt_widget.py:938-962)."""
        p = self.p
        spec = spectrum[:N_BARS]
        mx = float(np.max(spec)) if np.max(spec) > 0 else 1.0
        spec = spec / mx
        spec_draw = np.concatenate([spec, spec[::-1][1:-1]]) if p.radial_mirror else spec
        L = len(spec_draw)
        if L > 3 and p.radial_wave_smoothness > 0:
            spec_draw = smooth_closed(spec_draw, int(p.radial_wave_smoothness))
            L = len(spec_draw)
        if p.radial_temporal_alpha > 0.0:
            if self.prev is None or len(self.prev) != L:
                self.prev = spec_draw.copy()
            else:
                self.prev = p.radial_temporal_alpha * self.prev + (1.0 - p.radial_temporal_alpha) * spec_draw
            spec_draw = self.prev
        if L > 3:
            spec_draw = smooth_closed(spec_draw, int(p.radial_smooth_amount))
        return np.asarray(spec_draw)

    def path(self, spectrum, energy, w, h):
        """Vertices float64[L + 1, 2] of the closed path handed to ``p.drawPath`` (rt_widget.py:943-975)."""
        p = self.p
        spec_draw = self.values(spectrum)
        L = len(spec_draw)
        cx, cy = w / 2.0, h / 2.0
        inner = min(w, h) * 0.22
        span = min(w, h) * 0.28 * (0.6 + 0.6 * energy)
        out = np.empty((L + 1, 2), dtype=np.float64)
        for i in range(L + 1):
            a = 2.0 * np.pi * (i / L) - (np.pi / 2.0) + np.deg2rad(p.radial_rotation_deg)
            v = float(spec_draw[min(i, L - 1)])
            r = inner + span * v
            out[i, 0] = cx + np.cos(a) * r
            out[i, 1] = cy + np.sin(a) * r
        return out


def _fbm(theta, phase):
    v = 0.0
    amp = 1.0
    freq = 1.0
    for k in range(3):
        v += amp * np.sin(freq * theta + (k + 1) * phase)
        amp *= 0.5
        freq *= 2.0
    return v / 1.75


def _envelope_terms(spectrum, energy, w, h, p: RadialParams):
    noise_amt = float(p.edge_waviness) / 100.0
    audio_amt = float(p.feather_audio_amount) / 100.0 if p.feather_audio_enabled else 0.0
    s = np.asarray(spectrum, dtype=np.float32)
    mx = float(np.max(s)) if np.max(s) > 1e-9 else 1.0
    base_norm = s / mx
    spec_norm = np.concatenate([base_norm, base_norm[::-1][1:-1]]) if p.radial_mirror else base_norm
    amp_noise_px = min(w, h) * 0.025 * noise_amt
    energy_feather = min(1.0, energy * (p.feather_sensitivity / max(1e-3, p.waveform_sensitivity)))
    amp_audio_px = min(w, h) * 0.06 * audio_amt * (0.4 + 0.6 * energy_feather)
    return spec_norm, amp_noise_px, amp_audio_px


def envelope_path(spectrum, energy, phase, w, h, p: RadialParams | None = None):
    """Vertices float64[361, 2] of the clip / fill path of the centre disc (rt_widget.py:1007-1051)."""
    p = p or RadialParams()
    cx, cy = w / 2.0, h / 2.0
    inner = min(w, h) * 0.22
    spec_norm, amp_noise_px, amp_audio_px = _envelope_terms(spectrum, energy, w, h, p)
    N = N_ENV
    out = np.empty((N + 1, 2), dtype=np.float64)
    for i in range(N + 1):
        a = 2.0 * np.pi * (i / N) - (np.pi / 2.0) + np.deg2rad(p.radial_rotation_deg)
        si = int((i / N) * (len(spec_norm) - 1)) if len(spec_norm) > 1 else 0
        s_val = float(spec_norm[si]) if len(spec_norm) else 0.0
        r = inner + amp_noise_px * _fbm(a * 2.0, phase) + amp_audio_px * s_val
        r_max = 0.48 * min(w, h)
        if r_max > inner:
            g = max(0.0, (r - inner) / (r_max - inner))
            g = math.tanh(1.25 * g) / math.tanh(1.25)
            r = inner + g * (r_max - inner)
        out[i, 0] = cx + np.cos(a) * r
        out[i, 1] = cy + np.sin(a) * r
    return out


def feather_ring_paths(spectrum, energy, phase, w, h, p: RadialParams | None = None):
    """[(vertices float64[361, 2], alpha)] for k = 1..6 (rt_widget.py:1098-1117); empty unless feather_enabled."""
    p = p or RadialParams()
    if not p.feather_enabled:
        return []
    cx, cy = w / 2.0, h / 2.0
    inner = min(w, h) * 0.22
    spec_norm, amp_noise_px, amp_audio_px = _envelope_terms(spectrum, energy, w, h, p)
    N = N_ENV
    rings = []
    for k in range(1, 7):
        fscale = 1.0 + 0.015 * k
        out = np.empty((N + 1, 2), dtype=np.float64)
        for i in range(N + 1):
            a = 2.0 * np.pi * (i / N) - (np.pi / 2.0) + np.deg2rad(p.radial_rotation_deg)
            si = int((i / N) * (len(spec_norm) - 1)) if len(spec_norm) > 1 else 0
            s_val = float(spec_norm[si]) if len(spec_norm) else 0.0
            r = (inner + amp_noise_px * np.sin(a * 2.0 + phase) + amp_audio_px * s_val) * fscale
            out[i, 0] = cx + np.cos(a) * r
            out[i, 1] = cy + np.sin(a) * r
        rings.append((out, max(10, 70 - 10 * k)))
    return rings


def wave_circular_path(samples, energy, w, h, rotation_deg=0.0):
    """Vertices float64[n_vertices, 2] of the closed path of "Waveform - Circular" (rt_widget.py:867-896; export
    path: ``_offscreen_paint`` is True, so ``segs = min(720, n)`` whatever shadow / glow say)."""
    n = len(samples)
    cx, cy = w / 2.0, h / 2.0
    base_r = min(w, h) * 0.28
    amp = min(w, h) * 0.22 * (0.5 + 0.6 * energy)
    segs = min(720, n)
    step = max(1, int(n / segs))
    out = []
    for i in range(0, n, step):
        v = float(samples[i])
        a = 2.0 * np.pi * (i / max(1, n - 1)) - (np.pi / 2.0) + np.deg2rad(rotation_deg)
        r = base_r + v * amp
        out.append((cx + np.cos(a) * r, cy + np.sin(a) * r))
    return np.asarray(out, dtype=np.float64)
