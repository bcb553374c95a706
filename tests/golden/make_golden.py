#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference has no tests or golden files of its own, so these fixtures
are produced by importing its modules and driving them with seeded synthetic audio:

* ``audio.analysis.Analyzer``                       -- imported as is (numpy/scipy only);
* ``export.exporter.QPainterOffscreenRenderer``     -- imported with two stand-in modules for
  packages that are not installed here: ``soundfile`` (an in-memory array with
  seek/read/len/samplerate) and ``PySide6`` (a *recording* QPainter/QPainterPath/QPen/QColor that
  stores the arguments of every draw call instead of rasterising).  ``_read_frame`` and
  ``RTVisualizerWidget.paint_frame`` therefore execute the reference's own code; what is captured
  is (a) the analysis outputs and (b) the draw lists.  No pixel is produced: Qt's raster engine
  is not available, so pixel coverage stays "parity unpinned" (see oracle/raster_ref.py).

Usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
Fixtures are data only (inputs + expected outputs).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AURORA_REFERENCE", "/root/reference")

sys.path.insert(0, REPO)
from aurora_amd import synth  # noqa: E402  (input generator only)

# ----------------------------------------------------------------------------- stand-ins

_AUDIO = {}      # path -> (array [n] or [n, C] float32, samplerate)


class _SoundFile:
    def __init__(self, path, mode='r'):
        if path not in _AUDIO:
            raise RuntimeError(f"unknown in-memory audio {path!r}")
        self._data, self.samplerate = _AUDIO[path]
        self._pos = 0

    def __len__(self):
        return int(self._data.shape[0])

    def seek(self, frame):
        self._pos = int(frame)

    def read(self, frames, dtype='float32', always_2d=False):
        end = min(len(self), self._pos + int(frames))
        y = np.array(self._data[self._pos:end], dtype=np.float32, copy=True)
        self._pos = end
        return y


class QColor:
    HexArgb = 1

    def __init__(self, *a):
        self._valid = True
        if len(a) == 1 and isinstance(a[0], QColor):
            self.r, self.g, self.b, self.a = a[0].r, a[0].g, a[0].b, a[0].a
        elif len(a) == 1 and isinstance(a[0], str):
            s = a[0].lstrip('#')
            try:
                if len(s) == 8:
                    self.a, self.r, self.g, self.b = (int(s[i:i + 2], 16) for i in (0, 2, 4, 6))
                elif len(s) == 6:
                    self.r, self.g, self.b = (int(s[i:i + 2], 16) for i in (0, 2, 4))
                    self.a = 255
                else:
                    raise ValueError
            except ValueError:
                self.r = self.g = self.b = 0
                self.a = 255
                self._valid = False
        elif len(a) >= 3:
            self.r, self.g, self.b = int(a[0]), int(a[1]), int(a[2])
            self.a = int(a[3]) if len(a) > 3 else 255
        else:
            self.r = self.g = self.b = 0
            self.a = 255
            self._valid = False

    def red(self): return self.r
    def green(self): return self.g
    def blue(self): return self.b
    def alpha(self): return self.a
    # Qt 6 stores 16 bits per channel (a * 0x101) and alphaF() returns alpha / 65535 as a C float
    def alphaF(self): return float(np.float32(self.a * 257) / np.float32(65535))
    def isValid(self): return self._valid
    def setAlpha(self, a): self.a = int(a)
    def name(self, *_): return "#%02x%02x%02x%02x" % (self.a, self.r, self.g, self.b)
    def tup(self): return (self.r, self.g, self.b, self.a)


class QPen:
    def __init__(self, color=None):
        self.color = QColor(color) if color is not None else QColor(0, 0, 0)
        self.width = 1
        self.cap = "SquareCap"      # Qt default
        self.join = "BevelJoin"     # Qt default

    def setWidth(self, w): self.width = int(w)
    def setCapStyle(self, c): self.cap = c
    def setJoinStyle(self, j): self.join = j


class QPainterPath:
    def __init__(self):
        self.ops = []

    def moveTo(self, x, y): self.ops.append(("M", float(x), float(y)))
    def lineTo(self, x, y): self.ops.append(("L", float(x), float(y)))
    def closeSubpath(self): self.ops.append(("Z", 0.0, 0.0))

    def boundingRect(self):
        # a path of straight segments: Qt's bounding rect is the min / max of its points
        xs = [o[1] for o in self.ops if o[0] in "ML"]
        ys = [o[2] for o in self.ops if o[0] in "ML"]
        return QRectF(min(xs), min(ys), max(xs) - min(xs), max(ys) - min(ys))


class QRectF:
    """QRectF(x, y, w, h); adjusted() as qrect.h writes it: (xp + dx1, yp + dy1, w + dx2 - dx1, h + dy2 - dy1)."""

    def __init__(self, x, y, w, h):
        self._x, self._y, self._w, self._h = float(x), float(y), float(w), float(h)

    def adjusted(self, dx1, dy1, dx2, dy2):
        return QRectF(self._x + dx1, self._y + dy1, self._w + dx2 - dx1, self._h + dy2 - dy1)

    def x(self): return self._x
    def y(self): return self._y
    def width(self): return self._w
    def height(self): return self._h


class QSize:
    def __init__(self, w=0, h=0):
        self.w, self.h = int(w), int(h)

    def width(self): return self.w
    def height(self): return self.h


def _qsize_scaled(w, h, tw, th, aspect):
    """QSize(w, h).scaled(QSize(tw, th), aspect) (qsize.cpp) followed by QImage::scaled's clamp to 1x1."""
    if aspect == _Qt.IgnoreAspectRatio or w == 0 or h == 0:
        nw, nh = tw, th
    else:
        rw = (th * w) // h
        use_height = (rw <= tw) if aspect == _Qt.KeepAspectRatio else (rw >= tw)
        nw, nh = (rw, th) if use_height else (tw, (tw * h) // w)
    return max(nw, 1), max(nh, 1)


class QImage:
    """Recording image.  QImage(w, h, format) is a paint device; QImage(path) is "decoded" when the file name carries
    its size (``memimg_<w>x<h>``), null otherwise; scaled() records its arguments and returns an image of the size Qt
    would produce (no pixel is computed anywhere in this script)."""
    Format_RGB888 = 13

    def __init__(self, *a):
        self.args = a
        self.calls = []
        self.size = None
        self.scaled_from = None
        if len(a) == 1 and isinstance(a[0], str):
            import re
            m = re.search(r"memimg_(\d+)x(\d+)", a[0])
            if m:
                self.size = (int(m.group(1)), int(m.group(2)))
        elif len(a) >= 2:
            self.size = (int(a[0]), int(a[1]))

    def isNull(self): return self.size is None
    def width(self): return self.size[0]
    def height(self): return self.size[1]

    def scaled(self, *a):
        if isinstance(a[0], QSize):
            tw, th, aspect, mode = a[0].w, a[0].h, a[1], a[2]
        else:
            tw, th, aspect, mode = int(a[0]), int(a[1]), a[2], a[3]
        out = QImage(*_qsize_scaled(self.size[0], self.size[1], tw, th, aspect), 0)
        out.scaled_from = (self.size, (tw, th), int(aspect), int(mode))
        return out


class QPainter:
    Antialiasing = 1
    SmoothPixmapTransform = 2
    CompositionMode_Plus = 12
    CompositionMode_Multiply = 13

    def __init__(self, device=None):
        self.device = device
        self.calls = device.calls if device is not None else []
        self.hints = set()

    def isActive(self): return True
    def end(self): self.calls.append(("end",))

    def fillRect(self, *a):
        col = a[-1]
        self.calls.append(("fillRect", tuple(a[:-1]), col.tup()))

    def setPen(self, pen):
        if isinstance(pen, QColor):
            pen = QPen(pen)
        self.calls.append(("setPen", pen.color.tup(), pen.width, pen.cap, pen.join))

    def setRenderHint(self, hint, on=True):
        (self.hints.add if on else self.hints.discard)(hint)
        self.calls.append(("setRenderHint", hint, bool(on)))

    def drawLine(self, x1, y1, x2, y2): self.calls.append(("drawLine", x1, y1, x2, y2))
    def drawPoint(self, x, y): self.calls.append(("drawPoint", x, y))
    def drawPath(self, path): self.calls.append(("drawPath", list(path.ops)))

    def drawImage(self, x, y, img):
        self.calls.append(("drawImage", int(x), int(y), img.size, img.scaled_from))
    def fillPath(self, path, col): self.calls.append(("fillPath", list(path.ops), col.tup()))
    def setClipPath(self, path): self.calls.append(("setClipPath", list(path.ops)))
    def translate(self, dx, dy): self.calls.append(("translate", float(dx), float(dy)))
    def setCompositionMode(self, mode): self.calls.append(("setCompositionMode", int(mode)))
    def save(self): self.calls.append(("save",))
    def restore(self): self.calls.append(("restore",))

    def __getattr__(self, name):        # anything else: record the call
        def _rec(*a, **k):
            self.calls.append((name, "unrecorded-args"))
        return _rec


class _QWidget:
    def __init__(self, *a, **k): pass
    def update(self): pass
    def setMinimumSize(self, *a): pass
    def width(self): return 0
    def height(self): return 0
    def rect(self): return None


class _Qt:
    PreciseTimer = 0
    RoundCap = "RoundCap"
    RoundJoin = "RoundJoin"
    SquareCap = "SquareCap"
    KeepAspectRatio = 1
    IgnoreAspectRatio = 0
    KeepAspectRatioByExpanding = 2
    SmoothTransformation = 1


def _install_stand_ins():
    sf = types.ModuleType("soundfile")
    sf.SoundFile = _SoundFile
    sys.modules["soundfile"] = sf

    def _dummy(name):
        return type(name, (), {"__init__": lambda self, *a, **k: None})

    pkg = types.ModuleType("PySide6")
    pkg.__path__ = []
    core = types.ModuleType("PySide6.QtCore")
    core.Qt = _Qt
    for n in ("QTimer", "QRect", "QThread"):
        setattr(core, n, _dummy(n))
    core.QSize = QSize
    core.QCoreApplication = type("QCoreApplication", (), {"instance": staticmethod(lambda: None)})
    gui = types.ModuleType("PySide6.QtGui")
    gui.QPainter, gui.QColor, gui.QPen, gui.QImage, gui.QPainterPath = QPainter, QColor, QPen, QImage, QPainterPath
    for n in ("QPixmap", "QTransform", "QSurfaceFormat"):
        setattr(gui, n, _dummy(n))
    wid = types.ModuleType("PySide6.QtWidgets")
    wid.QWidget = _QWidget
    wid.QApplication = _dummy("QApplication")
    pkg.QtCore, pkg.QtGui, pkg.QtWidgets = core, gui, wid
    sys.modules.update({"PySide6": pkg, "PySide6.QtCore": core, "PySide6.QtGui": gui, "PySide6.QtWidgets": wid})


# ----------------------------------------------------------------------------- capture

def _track(kind, n, sr, channels, seed=1234):
    a = synth.make(kind, n, sr, channels=max(1, channels), seed=seed)
    return a[:, 0].copy() if channels == 0 else a          # channels == 0 -> 1-D mono file


def capture_analysis(name, kind, n, sr, fps, channels, sensitivity=1.0):
    """Reference _read_frame + Analyzer over a whole synthetic track."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track(kind, n, sr, channels)
    path = f"mem://{name}"
    _AUDIO[path] = (audio, sr)
    r = QPainterOffscreenRenderer(64, 64, "Spectrum - Linear", (0.2, 0.8, 1.0), sensitivity, fps, path)
    total = int(np.ceil(r.duration * fps))                    # exporter.py:536
    samples = np.zeros((total, 1024), np.float32)
    logspec = np.zeros((total, 1025), np.float32)
    flux = np.zeros(total, np.float64)
    energy = np.zeros(total, np.float64)
    pen = np.zeros((total, 3), np.uint8)
    phase = np.zeros(total, np.float64)
    v = r._view
    for i in range(total):
        t = i / fps                                           # exporter.py:539
        s, spec = r._read_frame(t)
        samples[i], logspec[i], flux[i] = s, spec, r._feeder._flux
        img = v.render_frame_to_qimage(64, 64, s, spec)       # runs paint_frame -> energy, colour, phase
        fill = [c for c in img.calls if c[0] == "fillRect"][0]
        assert fill[2][:3] == (13, 13, 18)
        sp = [c for c in img.calls if c[0] == "setPen"][0]
        pen[i] = sp[1][:3]
        phase[i] = v._phase
        energy[i] = float(np.clip(np.mean(np.abs(s)) * v.waveform_sensitivity, 0.0, 1.0))
    hop = max(1, int(sr / fps))
    np.savez_compressed(os.path.join(HERE, f"analysis_{name}.npz"), audio=audio, sample_rate=sr, fps=fps,
                        sensitivity=sensitivity, hop=hop, samples=samples, logspec=logspec, flux_smooth=flux,
                        energy=energy, pen_rgb=pen, phase=phase)
    print(f"analysis_{name}: {total} frames, hop {hop}, audio {audio.shape}")
    return audio


def _path_xy(ops):
    return np.array([(o[1], o[2]) for o in ops if o[0] in "ML"], dtype=np.float64)


def capture_drawlists(name, kind, n, sr, fps, sizes, frames):
    """Reference paint_frame draw calls for the three linear modes (and Radial path vertices)."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track(kind, n, sr, 2)
    path = f"mem://dl_{name}"
    _AUDIO[path] = (audio, sr)
    out = dict(audio=audio, sample_rate=sr, fps=fps, frames=np.asarray(frames), sizes=np.asarray(sizes))
    for mode, tag in (("Spectrum - Linear", "spec"), ("Waveform - Linear", "wave"), ("Particles", "part"),
                      ("Spectrum - Radial", "radial")):
        for (w, h) in sizes:
            r = QPainterOffscreenRenderer(w, h, mode, (0.2, 0.8, 1.0), 1.0, fps, path)
            total = int(np.ceil(r.duration * fps))
            keep = set(int(f) for f in frames)
            for i in range(total):
                s, spec = r._read_frame(i / fps)
                img = r._view.render_frame_to_qimage(w, h, s, spec)
                if i not in keep:
                    continue
                calls = img.calls
                key = f"{tag}_{w}x{h}_f{i}"
                sp = [c for c in calls if c[0] == "setPen"][0]
                assert sp[2] == 2 and sp[3] == "SquareCap" and sp[4] == "BevelJoin", sp
                assert not any(c[0] == "setRenderHint" for c in calls) or tag == "radial"
                out[key + "_pen"] = np.asarray(sp[1][:3], np.uint8)
                out[key + "_samples"] = s.copy()
                out[key + "_spectrum"] = spec.copy()
                if tag == "spec":
                    out[key + "_lines"] = np.asarray([c[1:] for c in calls if c[0] == "drawLine"], np.int64)
                elif tag == "part":
                    out[key + "_points"] = np.asarray([c[1:] for c in calls if c[0] == "drawPoint"], np.int64)
                elif tag == "wave":
                    dp = [c for c in calls if c[0] == "drawPath"]
                    assert len(dp) == 1
                    out[key + "_path"] = _path_xy(dp[0][1])
                else:
                    dp = [c for c in calls if c[0] == "drawPath"]
                    out[key + "_path"] = _path_xy(dp[0][1])
                    cp = [c for c in calls if c[0] == "setClipPath"]
                    out[key + "_clip"] = _path_xy(cp[0][1])
                    out[key + "_calls"] = np.asarray([c[0] for c in calls])
    np.savez_compressed(os.path.join(HERE, f"drawlists_{name}.npz"), **out)
    print(f"drawlists_{name}: {len(out)} arrays")


def capture_radial_view_state(name, kind, n, sr, fps, size, frames, view_state, sensitivity=1.0):
    """Spectrum - Radial with a non-default view_state: radial path, clip path, centre fill, feather rings."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track(kind, n, sr, 2)
    path = f"mem://rv_{name}"
    _AUDIO[path] = (audio, sr)
    w, h = size
    r = QPainterOffscreenRenderer(w, h, "Spectrum - Radial", (0.2, 0.8, 1.0), sensitivity, fps, path, **view_state)
    total = int(np.ceil(r.duration * fps))
    out = dict(audio=audio, sample_rate=sr, fps=fps, frames=np.asarray(frames), size=np.asarray(size),
               sensitivity=sensitivity, view_state_json=np.asarray(__import__("json").dumps(view_state)))
    keep = set(int(f) for f in frames)
    for i in range(total):
        s, spec = r._read_frame(i / fps)
        img = r._view.render_frame_to_qimage(w, h, s, spec)
        if i not in keep:
            continue
        calls = img.calls
        key = f"f{i}"
        out[key + "_path"] = _path_xy([c for c in calls if c[0] == "drawPath"][0][1])
        out[key + "_clip"] = _path_xy([c for c in calls if c[0] == "setClipPath"][0][1])
        fills = [c for c in calls if c[0] == "fillPath"]
        out[key + "_fill_rgba"] = np.asarray([c[2] for c in fills], np.int64)          # centre fill, then rings
        out[key + "_rings"] = np.asarray([_path_xy(c[1]) for c in fills[1:]], np.float64)
        out[key + "_calls"] = np.asarray([c[0] for c in calls])
        out[key + "_phase"] = np.float64(r._view._phase)
    np.savez_compressed(os.path.join(HERE, f"radialvs_{name}.npz"), **out)
    print(f"radialvs_{name}: {len(out)} arrays")


def capture_wave_circular(name, kind, n, sr, fps, sizes, frames, view_state, sensitivity=1.0):
    """Waveform - Circular (rt_widget.py:871-902): the closed path handed to drawPath, per frame size."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track(kind, n, sr, 2)
    path = f"mem://wc_{name}"
    _AUDIO[path] = (audio, sr)
    out = dict(audio=audio, sample_rate=sr, fps=fps, frames=np.asarray(frames), sizes=np.asarray(sizes),
               sensitivity=sensitivity, view_state_json=np.asarray(__import__("json").dumps(view_state)))
    keep = set(int(f) for f in frames)
    for (w, h) in sizes:
        r = QPainterOffscreenRenderer(w, h, "Waveform - Circular", (0.2, 0.8, 1.0), sensitivity, fps, path,
                                      **view_state)
        total = int(np.ceil(r.duration * fps))
        for i in range(total):
            s, spec = r._read_frame(i / fps)
            img = r._view.render_frame_to_qimage(w, h, s, spec)
            if i not in keep:
                continue
            calls = img.calls
            key = f"{w}x{h}_f{i}"
            sp = [c for c in calls if c[0] == "setPen"][0]
            assert sp[2] == 2 and sp[3] == "SquareCap" and sp[4] == "BevelJoin", sp
            dp = [c for c in calls if c[0] == "drawPath"]
            assert len(dp) == 1 and dp[0][1][-1][0] == "Z"
            out[key + "_pen"] = np.asarray(sp[1][:3], np.uint8)
            out[key + "_path"] = _path_xy(dp[0][1])
            out[key + "_calls"] = np.asarray([c[0] for c in calls])
    np.savez_compressed(os.path.join(HERE, f"wavecirc_{name}.npz"), **out)
    print(f"wavecirc_{name}: {len(out)} arrays")


def capture_fill(name, mode, kind, n, sr, fps, size, view_state, sensitivity=1.0):
    """`_apply_fill` (rt_widget.py:856-864): for EVERY frame whether fillPath was called, its colour, the composition
    mode set before it and the position of the call among the others; fill path vertices of a few frames."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track(kind, n, sr, 2)
    path = f"mem://fill_{name}"
    _AUDIO[path] = (audio, sr)
    w, h = size
    r = QPainterOffscreenRenderer(w, h, mode, (0.2, 0.8, 1.0), sensitivity, fps, path, **view_state)
    total = int(np.ceil(r.duration * fps))
    filled = np.zeros(total, np.uint8)
    comp = np.full(total, -1, np.int64)
    rgba = np.zeros((total, 4), np.int64)
    out = dict(audio=audio, sample_rate=sr, fps=fps, size=np.asarray(size), sensitivity=sensitivity, mode=np.asarray(mode),
               view_state_json=np.asarray(__import__("json").dumps(view_state)))
    for i in range(total):
        s, spec = r._read_frame(i / fps)
        calls = r._view.render_frame_to_qimage(w, h, s, spec).calls
        names = [c[0] for c in calls]
        fills = [c for c in calls if c[0] == "fillPath" and c[2][:3] != (30, 30, 40)]
        if fills:
            assert len(fills) == 1
            filled[i] = 1
            rgba[i] = fills[0][2]
            k = names.index("fillPath")
            assert k < names.index("drawPath")                       # the fill comes before the stroke
            cm = [c for c in calls[:k] if c[0] == "setCompositionMode"]
            comp[i] = cm[-1][1] if cm else -1
            dp = [c for c in calls if c[0] == "drawPath"][0]
            assert fills[0][1] == dp[1]                               # same path object
            if i < 6:
                out[f"f{i}_path"] = _path_xy(fills[0][1])
    out.update(filled=filled, comp=comp, rgba=rgba)
    np.savez_compressed(os.path.join(HERE, f"fill_{name}.npz"), **out)
    print(f"fill_{name}: {int(filled.sum())}/{total} frames filled, rgba {rgba[filled > 0][:1].tolist()}")


def capture_shadow(name, mode, kind, n, sr, fps, size, view_state, frames):
    """`_draw_shadow_path` (rt_widget.py:579-594): translate arguments, shadow pen and the order of the calls."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track(kind, n, sr, 2)
    path = f"mem://shadow_{name}"
    _AUDIO[path] = (audio, sr)
    w, h = size
    r = QPainterOffscreenRenderer(w, h, mode, (0.2, 0.8, 1.0), 1.0, fps, path, **view_state)
    total = int(np.ceil(r.duration * fps))
    out = dict(audio=audio, sample_rate=sr, fps=fps, size=np.asarray(size), mode=np.asarray(mode),
               frames=np.asarray(frames), view_state_json=np.asarray(__import__("json").dumps(view_state)))
    keep = set(int(f) for f in frames)
    for i in range(total):
        s, spec = r._read_frame(i / fps)
        calls = r._view.render_frame_to_qimage(w, h, s, spec).calls
        if i not in keep:
            continue
        out[f"f{i}_calls"] = np.asarray([c[0] for c in calls])
        out[f"f{i}_translate"] = np.asarray([c[1:] for c in calls if c[0] == "translate"], np.float64)
        pens = [c for c in calls if c[0] == "setPen"]
        out[f"f{i}_pens_rgba"] = np.asarray([c[1] for c in pens], np.int64)
        out[f"f{i}_pens_width"] = np.asarray([c[2] for c in pens], np.int64)
        out[f"f{i}_pens_capjoin"] = np.asarray([f"{c[3]}/{c[4]}" for c in pens])
        dps = [c for c in calls if c[0] == "drawPath"]
        out[f"f{i}_n_drawpath"] = np.int64(len(dps))
        out[f"f{i}_paths_equal"] = np.asarray([dps[k][1] == dps[k + 1][1] for k in range(len(dps) - 1)])
        out[f"f{i}_path_lens"] = np.asarray([len(d[1]) for d in dps], np.int64)
    np.savez_compressed(os.path.join(HERE, f"shadow_{name}.npz"), **out)
    print(f"shadow_{name}: {len(out)} arrays;", list(out[f"f{frames[0]}_calls"]))


def capture_glow(name, mode, kind, n, sr, fps, size, view_state, frames, sensitivity=1.0):
    """`_draw_glow_path` -> `_draw_glow_offscreen` (rt_widget.py:598-616, :634-653): the pens of every frame (colour with
    its energy-dependent alpha, width, cap / join), composition mode, render hint and the order of the calls."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track(kind, n, sr, 2)
    path = f"mem://glow_{name}"
    _AUDIO[path] = (audio, sr)
    w, h = size
    r = QPainterOffscreenRenderer(w, h, mode, (0.2, 0.8, 1.0), sensitivity, fps, path, **view_state)
    total = int(np.ceil(r.duration * fps))
    out = dict(audio=audio, sample_rate=sr, fps=fps, size=np.asarray(size), mode=np.asarray(mode),
               sensitivity=np.float64(sensitivity), frames=np.asarray(frames),
               view_state_json=np.asarray(__import__("json").dumps(view_state)))
    keep = set(int(f) for f in frames)
    pens_rgba = np.zeros((total, 16, 4), np.int64)      # glow pens of every frame in paint order, zero padded
    pens_width = np.zeros((total, 16), np.int64)
    n_pens = np.zeros(total, np.int64)
    for i in range(total):
        s, spec = r._read_frame(i / fps)
        calls = r._view.render_frame_to_qimage(w, h, s, spec).calls
        plus, glow_pens, glow_paths, hints = False, [], 0, []
        for c in calls:
            if c[0] == "setCompositionMode":
                plus = c[1] == QPainter.CompositionMode_Plus
            elif c[0] == "restore":
                plus = False
            elif plus and c[0] == "setPen":
                glow_pens.append(c)
                assert (c[3], c[4]) == ("RoundCap", "RoundJoin")
            elif plus and c[0] == "drawPath":
                glow_paths += 1
            elif plus and c[0] == "setRenderHint":
                hints.append(c[1:])
        assert glow_paths == len(glow_pens)
        n_pens[i] = len(glow_pens)
        for k, c in enumerate(glow_pens):
            pens_rgba[i, k], pens_width[i, k] = c[1], c[2]
        if i in keep:
            out[f"f{i}_calls"] = np.asarray([c[0] for c in calls])
            dps = [c for c in calls if c[0] == "drawPath"]
            out[f"f{i}_paths_equal"] = np.asarray([dps[k][1] == dps[k + 1][1] for k in range(len(dps) - 1)])
            out[f"f{i}_path"] = _path_xy(dps[-1][1])
            out[f"f{i}_comp_modes"] = np.asarray([c[1] for c in calls if c[0] == "setCompositionMode"], np.int64)
    out.update(pens_rgba=pens_rgba, pens_width=pens_width, n_pens=n_pens)
    np.savez_compressed(os.path.join(HERE, f"glow_{name}.npz"), **out)
    print(f"glow_{name}: pens per frame {n_pens.min()}..{n_pens.max()}, widths {pens_width[0][:int(n_pens[0])].tolist()},",
          f"alpha range {pens_rgba[:, 0, 3].min()}..{pens_rgba[:, 0, 3].max()};", list(out[f"f{frames[0]}_calls"]))


# the case matrix of the FFmpeg argv fixture; tests/test_host_cpu.py replays it against aurora_amd.export.exporter
FFMPEG_SYSTEMS = ["Linux", "Darwin", "Windows"]
FFMPEG_ENCODER_SETS = [
    ["libx264"], ["h264_nvenc", "hevc_nvenc"], ["h264_nvenc"], ["hevc_amf"], ["h264_amf", "hevc_amf"], ["h264_qsv"],
    ["h264_vaapi", "hevc_vaapi"], ["h264_videotoolbox", "hevc_videotoolbox"],
    ["libx264", "h264_nvenc", "hevc_nvenc", "h264_amf", "hevc_amf", "h264_qsv", "hevc_qsv", "h264_vaapi", "hevc_vaapi",
     "h264_videotoolbox", "hevc_videotoolbox"],
]
FFMPEG_DEVICES = [None, "", "cpu", "off", "0", "auto", " AUTO ", "nvenc:1", "nvenc:x", "nvenc", "amf:auto", "qsv:auto",
                  "vaapi:/dev/dri/renderD128", "videotoolbox", "videotoolbox:0", "bogus"]


def capture_center_image(name, kind, n, sr, fps, size, frames, image_size, view_state, sensitivity=1.0):
    """Spectrum - Radial with a centre image (rt_widget.py:1060-1090): per frame the clip path, the arguments of
    `img.scaled(out_w, out_h, IgnoreAspectRatio, SmoothTransformation)` and of `p.drawImage(dx, dy, ...)`."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track(kind, n, sr, 2)
    path = f"mem://ci_{name}"
    _AUDIO[path] = (audio, sr)
    w, h = size
    vs = dict(view_state, center_image_path=f"/nonexistent/memimg_{image_size[0]}x{image_size[1]}.png")
    r = QPainterOffscreenRenderer(w, h, "Spectrum - Radial", (0.2, 0.8, 1.0), sensitivity, fps, path, **vs)
    assert r._view.center_image is not None and not r._view.center_image.isNull()
    total = int(np.ceil(r.duration * fps))
    out = dict(audio=audio, sample_rate=sr, fps=fps, frames=np.asarray(frames), size=np.asarray(size),
               image_size=np.asarray(image_size), sensitivity=sensitivity,
               view_state_json=np.asarray(__import__("json").dumps(view_state)))
    keep = set(int(f) for f in frames)
    for i in range(total):
        s, spec = r._read_frame(i / fps)
        img = r._view.render_frame_to_qimage(w, h, s, spec)
        if i not in keep:
            continue
        calls = img.calls
        di = [c for c in calls if c[0] == "drawImage"]
        assert len(di) == 1 and not any(c[0] == "fillPath" for c in calls)
        _, dx, dy, out_size, sf = di[0]
        assert sf[0] == tuple(image_size) and sf[1] == out_size and sf[2] == _Qt.IgnoreAspectRatio and sf[3] == 1
        names = [c[0] for c in calls]
        k = names.index("setClipPath")
        assert ("setRenderHint", QPainter.Antialiasing, True) in calls[:k] and names.index("drawImage") > k
        key = f"f{i}"
        out[key + "_clip"] = _path_xy(calls[k][1])
        out[key + "_placement"] = np.asarray([out_size[0], out_size[1], dx, dy], np.int64)
        out[key + "_phase"] = np.float64(r._view._phase)
        out[key + "_energy"] = np.float64(np.clip(np.mean(np.abs(s)) * r._view.waveform_sensitivity, 0.0, 1.0))
    np.savez_compressed(os.path.join(HERE, f"images_center_{name}.npz"), **out)
    print(f"images_center_{name}: {len(out)} arrays")


def capture_background(name, cases):
    """`_draw_background` (rt_widget.py:505-575) for (frame size, image size, scale mode, offsets, dim) cases: every
    `drawImage(x, y, image)` with the size of what is drawn and the `scaled()` arguments behind it, and the dim fill."""
    from export.exporter import QPainterOffscreenRenderer
    audio = _track("pink", 2000, 48000, 2)
    path = f"mem://bg_{name}"
    _AUDIO[path] = (audio, 48000)
    rows, draws = [], []
    for (w, h, iw, ih, mode, ox, oy, dim) in cases:
        vs = dict(background_path=f"/nonexistent/memimg_{iw}x{ih}.png", background_scale_mode=mode,
                  background_offset_x=ox, background_offset_y=oy, background_dim_percent=dim)
        r = QPainterOffscreenRenderer(w, h, "Spectrum - Linear", (0.2, 0.8, 1.0), 1.0, 60, path, **vs)
        s, spec = r._read_frame(0.0)
        calls = r._view.render_frame_to_qimage(w, h, s, spec).calls
        names = [c[0] for c in calls]
        assert names[0] == "fillRect" and calls[0][2][:3] == (13, 13, 18)
        first_pen = names.index("setPen")
        di = [c for c in calls[:first_pen] if c[0] == "drawImage"]
        dims = [c for c in calls[1:first_pen] if c[0] == "fillRect"]
        assert len(dims) <= 1 and all(c[1] == (0, 0, w, h) and c[2][:3] == (0, 0, 0) for c in dims)
        dim_alpha = dims[0][2][3] if dims else 0
        for c in di:
            sf = c[4]
            draws.append([len(rows), c[1], c[2], c[3][0], c[3][1], -1 if sf is None else sf[2]])
            if sf is not None:
                assert sf[0] == (iw, ih) and sf[1] == (w, h) and sf[3] == 1
        rows.append([w, h, iw, ih, ox, oy, dim, dim_alpha, len(di)])
    np.savez_compressed(os.path.join(HERE, f"images_background_{name}.npz"),
                        cases=np.asarray(rows, np.int64), modes=np.asarray([c[4] for c in cases]),
                        draws=np.asarray(draws, np.int64))
    print(f"images_background_{name}: {len(rows)} cases, {len(draws)} drawImage calls")


class _FakeDri:
    """Stands in for pathlib.Path inside the exporter module: /dev/dri with two render nodes."""
    def __init__(self, p): self.p = str(p)
    def exists(self): return True
    def glob(self, pat): return [_FakeDri("/dev/dri/renderD129"), _FakeDri("/dev/dri/renderD128")]
    def __lt__(self, o): return self.p < o.p
    def __str__(self): return self.p
    @property
    def name(self): return self.p.rsplit("/", 1)[-1]


def replay_ffmpeg_cases(mod):
    """Runs build_ffmpeg_cmd / list_gpu_export_devices of ``mod`` (the reference's export.exporter here, the HIP
    build's in the test) over the case matrix with the machine probes replaced: FFmpeg binary, its encoder list,
    platform.system, nvidia-smi lookup and the /dev/dri listing."""
    import json
    saved = (mod._ffmpeg_bin, mod._ffmpeg_encoders, mod.platform.system, mod.shutil.which, mod.Path)
    out = {}
    try:
        mod._ffmpeg_bin = lambda: "/opt/bin/ffmpeg"
        mod.shutil.which = lambda name: None
        mod.Path = _FakeDri
        for system in FFMPEG_SYSTEMS:
            mod.platform.system = lambda system=system: system
            for k, encs in enumerate(FFMPEG_ENCODER_SETS):
                mod._ffmpeg_encoders = lambda _bin, encs=encs: set(encs)
                out[f"list/{system}/{k}"] = mod.list_gpu_export_devices()
                for j, dev in enumerate(FFMPEG_DEVICES):
                    for hevc in (False, True):
                        outp = "clip" if (j + hevc) % 2 else "clip.mkv"
                        try:
                            cmd, fmt = mod.build_ffmpeg_cmd(1920, 1080, 60, outp, "in.wav", prefer_hevc=hevc,
                                                            gpu_device=dev)
                            res = {"cmd": list(cmd), "fmt": fmt}
                        except RuntimeError as exc:
                            res = {"error": str(exc)}
                        out[f"cmd/{system}/{k}/{j}/{int(hevc)}"] = res
    finally:
        mod._ffmpeg_bin, mod._ffmpeg_encoders, mod.platform.system, mod.shutil.which, mod.Path = saved
    return json.dumps(out, sort_keys=True)


def capture_ffmpeg_cmds():
    """`build_ffmpeg_cmd` / `list_gpu_export_devices` (export/exporter.py:368-491) over the case matrix."""
    import export.exporter as ref
    js = replay_ffmpeg_cases(ref)
    np.savez_compressed(os.path.join(HERE, "ffmpeg_cmds.npz"), cases_json=np.asarray(js))
    print(f"ffmpeg_cmds: {js.count('cmd/')} argv cases, {len(js)} bytes of JSON")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ffmpeg":         # adds the ffmpeg_cmds fixture only
        _install_stand_ins()
        sys.path.insert(0, REF)
        capture_ffmpeg_cmds()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "glow":           # adds the glow_* fixtures only
        _install_stand_ins()
        sys.path.insert(0, REF)
        capture_glow("radial", "Spectrum - Radial", "pink", 24000, 48000, 60, (640, 368), dict(glow_enabled=True), [0, 3])
        capture_glow("circular", "Waveform - Circular", "sweep", 24000, 48000, 60, (640, 368),
                     dict(glow_enabled=True, glow_color="#FFFF4060", glow_radius=9, glow_strength=100,
                          shadow_enabled=True, radial_fill_enabled=True, radial_fill_threshold=0.0), [0, 3],
                     sensitivity=2.5)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "images":         # adds the images_* fixtures only
        _install_stand_ins()
        sys.path.insert(0, REF)
        capture_center_image("static", "pink", 6000, 48000, 60, (1280, 720), [0, 1, 4, 7], (640, 480), {})
        capture_center_image("motion_wavy", "pink", 12000, 48000, 60, (1920, 1080), [0, 1, 3, 8, 14], (333, 517),
                             dict(center_motion=70, center_image_zoom=140, edge_waviness=60, feather_audio_enabled=True,
                                  feather_audio_amount=80), sensitivity=1.8)
        capture_center_image("small_zoomed_out", "sweep", 11025, 44100, 30, (1000, 562), [0, 2, 5], (4000, 3000),
                             dict(center_motion=100, center_image_zoom=50, radial_rotation_deg=25.0))
        cases = []
        for (w, h) in ((1280, 720), (1000, 562), (3840, 2160)):
            for (iw, ih) in ((640, 480), (4000, 1500), (300, 900), (1280, 720)):
                for mode in ("fill", "fit", "stretch", "center", "tile"):
                    cases.append((w, h, iw, ih, mode, 0, 0, 0))
        cases += [(1280, 720, 640, 480, "fill", 25, -40, 35), (1280, 720, 640, 480, "fit", -300, 10, 100),
                  (1280, 720, 500, 300, "tile", 37, 21, 50), (1280, 720, 500, 300, "tile", -37, -21, 1),
                  (1920, 1080, 800, 600, "center", 111, -77, 99), (1920, 1080, 7, 5, "FILL", 0, 0, 150),
                  (1920, 1080, 800, 600, "unknown-mode", 3, 4, -5)]
        capture_background("plans", cases)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "multichannel":   # adds the analysis_*ch* fixtures only
        _install_stand_ins()
        sys.path.insert(0, REF)
        # y.mean(axis=1) over more than two channels (exporter.py:233-234): 5.1 and 3-channel tracks, 9 channels for
        # numpy's 8-accumulator pairwise order
        capture_analysis("pink_48k_60_6ch", "pink", 20000, 48000, 60, 6)
        capture_analysis("pink_44k1_30_3ch", "pink", 15000, 44100, 30, 3, sensitivity=1.5)
        capture_analysis("pink_48k_60_9ch", "pink", 9000, 48000, 60, 9)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "shadow":         # adds the shadow_* fixtures only
        _install_stand_ins()
        sys.path.insert(0, REF)
        capture_shadow("radial", "Spectrum - Radial", "pink", 6000, 48000, 60, (640, 368),
                       dict(shadow_enabled=True), [0, 3])
        capture_shadow("circular", "Waveform - Circular", "pink", 6000, 48000, 60, (640, 368),
                       dict(shadow_enabled=True, shadow_opacity=35, shadow_blur_radius=5, shadow_distance=11,
                            shadow_angle_deg=200, radial_fill_enabled=True, radial_fill_threshold=0.0), [0, 3])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "shadow_linear":  # adds the shadow_linear fixture only
        _install_stand_ins()
        sys.path.insert(0, REF)
        capture_shadow("linear", "Waveform - Linear", "pink", 6000, 48000, 60, (640, 368),
                       dict(shadow_enabled=True, shadow_opacity=50, shadow_blur_radius=6, shadow_distance=9,
                            shadow_angle_deg=120), [0, 3])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fill":           # adds the fill_* fixtures only
        _install_stand_ins()
        sys.path.insert(0, REF)
        # thresholds inside the energy range of the track, so that some frames are filled and some are not
        capture_fill("radial_normal", "Spectrum - Radial", "pink", 24000, 48000, 60, (640, 368),
                     dict(radial_fill_enabled=True, radial_fill_threshold=0.087))
        capture_fill("circular_add", "Waveform - Circular", "pink", 24000, 48000, 60, (640, 368),
                     dict(radial_fill_enabled=True, radial_fill_color="#6040C0FF", radial_fill_blend="add",
                          radial_fill_threshold=0.07), sensitivity=0.8)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "wave_circular":  # adds the wavecirc_* fixtures only
        _install_stand_ins()
        sys.path.insert(0, REF)
        capture_wave_circular("pink_48k_60", "pink", 12000, 48000, 60, [(1280, 720), (3840, 2160)], [0, 1, 5, 14], {})
        capture_wave_circular("sweep_44k1_30_rot", "sweep", 22050, 44100, 30, [(1000, 562)], [0, 3, 7],
                              dict(radial_rotation_deg=33.0), sensitivity=1.7)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "radial_vs":      # adds the radialvs_* fixtures only
        _install_stand_ins()
        sys.path.insert(0, REF)
        capture_radial_view_state("mirror_wavy", "pink", 12000, 48000, 60, (1280, 720), [0, 1, 4, 9, 14],
                                  dict(radial_mirror=True, edge_waviness=40, feather_enabled=True,
                                       feather_audio_enabled=True, feather_audio_amount=60,
                                       radial_rotation_deg=17.5, radial_smooth_amount=30,
                                       radial_waveform_smoothness=80, radial_temporal_smoothing=60))
        capture_radial_view_state("plain_nosmooth", "sweep", 22050, 44100, 30, (1000, 562), [0, 2, 7],
                                  dict(radial_smooth_amount=0, radial_waveform_smoothness=0,
                                       radial_temporal_smoothing=0, edge_waviness=100), sensitivity=2.0)
        return
    if not os.path.isdir(REF):
        raise SystemExit(f"reference not found at {REF}; fixtures can only be regenerated in the build container")
    _install_stand_ins()
    sys.path.insert(0, REF)
    # analysis fixtures: the three hop regimes of SURVEY 8(c) (+ mono, + ragged EOF lengths)
    capture_analysis("sweep_44k1_30", "sweep", 44100 + 777, 44100, 30, 2)       # hop 1470 > 1024
    capture_analysis("pink_48k_60", "pink", 47000, 48000, 60, 2)                # hop 800 (+224 zeros), short last read
    capture_analysis("pink_44k1_60", "pink", 22050 + 123, 44100, 60, 2)         # hop 735 (odd)
    capture_analysis("sweep_48k_60_mono", "sweep", 24000 + 5, 48000, 60, 0)     # 1-D mono file
    capture_analysis("pink_48k_24_sens", "pink", 30000, 48000, 24, 2, sensitivity=2.5)   # hop 2000, sensitivity != 1
    capture_analysis("tiny_48k_60", "pink", 500, 48000, 60, 2)                  # shorter than one hop
    capture_drawlists("pink_48k_60", "pink", 12000, 48000, 60,
                      sizes=[(1280, 720), (1920, 1080), (3840, 2160)], frames=[0, 1, 2, 5, 9, 13, 14])
    capture_drawlists("sweep_44k1_30", "sweep", 22050, 44100, 30,
                      sizes=[(1280, 720), (1000, 562)], frames=[0, 3, 7, 14])


if __name__ == "__main__":
    main()
