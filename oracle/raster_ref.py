"""ORACLE (test infrastructure, not product code): CPU restatement of Aurora's export rasteriser.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``aurora_amd/`` imports it.

Two layers:

1. **Draw lists** -- the arguments the reference hands to QPainter, restated from
   ``widgets/rt_widget.py``: ``drawlist_spectrum_linear`` (:934-940, :982-988),
   ``drawlist_waveform_linear`` (:903-913), ``drawlist_particles`` (:990-1000).  These ARE pinned:
   ``tests/golden/drawlists_*.npz`` were captured by driving the reference's own ``paint_frame``
   with a recording painter (``tests/golden/make_golden.py``).

2. **Coverage** -- which pixels a width-2, SquareCap, BevelJoin, non-antialiased QPen lights for those
   draw calls.  Qt's raster engine is not part of the reference tree and the reference has no frame
   goldens, so this layer is **parity unpinned** against Qt.  The rule implemented (and mirrored
   bit-for-bit by the HIP kernels) is:

   * a pixel is lit iff its centre ``(px + 0.5, py + 0.5)`` lies inside the stroke outline (closed set);
   * stroke outline of a polyline = union of one rectangle per segment (half-width 1, extended by 1 px
     beyond the first and last vertex = SquareCap) and one triangle per interior vertex on the outer
     side of the turn (BevelJoin);
   * integer primitives follow from that: ``drawLine(x0, ya, x0, yb)`` lights columns ``{x0-1, x0}``,
     rows ``min(ya,yb)-1 .. max(ya,yb)``; ``drawPoint(x, y)`` lights the 2x2 block
     ``{x-1, x} x {y-1, y}``; everything is clipped to the image;
   * background is ``fillRect(0,0,w,h,(13,13,18))`` (rt_widget.py:829), pen colour is opaque.

   Geometry is evaluated in float64 with the operation order below; the GPU kernels use the same
   order with contraction disabled, so GPU == oracle is a bit-exact contract.

Frames are represented through **column spans**: ``spans[W, K, 2]`` (int32) holds up to K half-open
row intervals ``[r0, r1)`` per pixel column.
"""
from __future__ import annotations

import math

import numpy as np

BG = (13, 13, 18)           # rt_widget.py:829
N_BARS = 80                 # rt_widget.py:935
N_PARTICLES = 50            # rt_widget.py:991
K_SPECTRUM = 2
K_PARTICLES = 2
K_WAVEFORM = 5
_BIG = 1.0e6


# --------------------------------------------------------------------------- draw lists

def spectrum_bar_heights(spectrum: np.ndarray, energy: float, h: int) -> np.ndarray:
    """rt_widget.py:938-940, :985-986 -> bh[80] (python ints)."""
    spec = spectrum[:N_BARS]
    mx = float(np.max(spec)) if np.max(spec) > 0 else 1.0
    spec = spec / mx
    bh = np.empty(N_BARS, dtype=np.int64)
    for i in range(N_BARS):
        v = float(spec[i])
        bh[i] = int(v * (h - 10) * (0.4 + 0.6 * energy))
    return bh


def drawlist_spectrum_linear(spectrum, energy, w, h):
    """-> int64[80,4] rows (x0, h-5, x0, h-5-bh) as passed to ``p.drawLine`` (rt_widget.py:983-988)."""
    bh = spectrum_bar_heights(spectrum, energy, h)
    bar_w = max(1, int(w / N_BARS))
    out = np.empty((N_BARS, 4), dtype=np.int64)
    for i in range(N_BARS):
        x0 = i * bar_w
        out[i] = (x0, h - 5, x0, h - 5 - int(bh[i]))
    return out


def waveform_col2samp(w: int, n: int = 1024) -> np.ndarray:
    """rt_widget.py:908 ``i = int((x / max(1, w - 1)) * (n - 1))`` for every column (index map)."""
    out = np.empty(w, dtype=np.int32)
    for x in range(w):
        out[x] = int((x / max(1, w - 1)) * (n - 1))
    return out


def drawlist_waveform_linear(samples, energy, w, h):
    """-> float64[w] path y per integer x (rt_widget.py:904-913).

    ``samples[i] * amp`` is np.float32 * python float -> float32 under NEP 50, and
    ``mid - <float32>`` stays float32, so y is a float32 value widened by ``float()``.
    """
    n = len(samples)
    amp = 0.45 * (h / 2) * (0.5 + energy)
    mid = h / 2.0
    idx = waveform_col2samp(w, n)
    ys = np.empty(w, dtype=np.float64)
    for x in range(w):
        ys[x] = float(mid - samples[idx[x]] * amp)
    return ys


def drawlist_particles(spectrum, w, h):
    """-> int64[50,2] (x, y) as passed to ``p.drawPoint`` (rt_widget.py:994-1000)."""
    spec = spectrum[:N_PARTICLES]
    spec = spec / (np.max(spec) + 1e-6)
    out = np.empty((N_PARTICLES, 2), dtype=np.int64)
    for i in range(N_PARTICLES):
        v = float(spec[i])
        x = int((i + 0.5) / N_PARTICLES * w)
        y = int(h / 2 + (v - 0.5) * h * 0.6)
        out[i] = (x, y)
    return out


# --------------------------------------------------------------------------- spans

def _clip_rows(r0, r1, h):
    r0 = np.clip(r0, 0, h)
    r1 = np.clip(r1, 0, h)
    empty = r1 <= r0
    r0 = np.where(empty, 0, r0)
    r1 = np.where(empty, 0, r1)
    return r0, r1


def spans_spectrum_linear(spectrum, energy, w, h):
    """Column spans of the 80 bars: slot 0 = bar whose x0 == px, slot 1 = bar whose x0 - 1 == px."""
    dl = drawlist_spectrum_linear(spectrum, energy, w, h)
    spans = np.zeros((w, K_SPECTRUM, 2), dtype=np.int32)
    for i in range(N_BARS):
        x0, ya, _, yb = (int(v) for v in dl[i])
        r0 = min(ya, yb) - 1
        r1 = max(ya, yb) + 1
        r0 = max(0, min(h, r0))
        r1 = max(0, min(h, r1))
        if r1 <= r0:
            r0 = r1 = 0
        if 0 <= x0 < w:
            spans[x0, 0] = (r0, r1)
        if 0 <= x0 - 1 < w:
            spans[x0 - 1, 1] = (r0, r1)
    return spans


def spans_particles(spectrum, w, h):
    """Column spans of the 50 points: slot 0 = point whose x == px, slot 1 = point whose x - 1 == px."""
    dl = drawlist_particles(spectrum, w, h)
    spans = np.zeros((w, K_PARTICLES, 2), dtype=np.int32)
    for i in range(N_PARTICLES):
        x, y = int(dl[i, 0]), int(dl[i, 1])
        r0 = max(0, min(h, y - 1))
        r1 = max(0, min(h, y + 1))
        if r1 <= r0:
            r0 = r1 = 0
        if 0 <= x < w:
            spans[x, 0] = (r0, r1)
        if 0 <= x - 1 < w:
            spans[x - 1, 1] = (r0, r1)
    return spans


def _rows_from_interval(ylo, yhi, valid, h):
    """Rows whose centre py + 0.5 lies in [ylo, yhi]; float64, clamped before the int conversion."""
    ylo = np.minimum(np.maximum(ylo, -_BIG), _BIG)
    yhi = np.minimum(np.maximum(yhi, -_BIG), _BIG)
    r0 = np.ceil(ylo - 0.5).astype(np.int64)
    r1 = np.floor(yhi - 0.5).astype(np.int64) + 1
    r0, r1 = _clip_rows(r0, r1, h)
    r0 = np.where(valid, r0, 0)
    r1 = np.where(valid, r1, 0)
    empty = r1 <= r0
    return np.where(empty, 0, r0).astype(np.int32), np.where(empty, 0, r1).astype(np.int32)


def _segment_interval(Y, seg, px, w):
    """y-interval of segment ``seg``'s stroke rectangle on the vertical line x = px + 0.5.

    seg, px: int arrays of equal shape.  Returns (ylo, yhi, valid).
    """
    nseg = w - 1
    ok = (seg >= 0) & (seg < nseg)
    s = np.clip(seg, 0, max(0, nseg - 1))
    y0 = Y[s]
    y1 = Y[np.minimum(s + 1, w - 1)]
    dy = y1 - y0
    L2 = 1.0 + dy * dy
    L = np.sqrt(L2)
    rx = (px + 0.5) - s
    c = rx * dy
    tlo = c - L
    thi = c + L
    s0L = np.where(s == 0, -L, 0.0)
    s1L = np.where(s == nseg - 1, L2 + L, L2)
    with np.errstate(divide='ignore', invalid='ignore'):
        qa = (s0L - rx) / dy
        qb = (s1L - rx) / dy
    pos = dy > 0.0
    neg = dy < 0.0
    flat = ~(pos | neg)
    slo = np.where(pos, qa, np.where(neg, qb, -_BIG))
    shi = np.where(pos, qb, np.where(neg, qa, _BIG))
    flat_in = (s0L <= rx) & (rx <= s1L)
    ok = ok & (~flat | flat_in)
    lo = np.maximum(tlo, slo)
    hi = np.minimum(thi, shi)
    return y0 + lo, y0 + hi, ok


def _bevel_interval(Y, vtx, px, w):
    """y-interval of the BevelJoin triangle at interior vertex ``vtx`` on the line x = px + 0.5."""
    ok = (vtx >= 1) & (vtx <= w - 2)
    v = np.clip(vtx, 1, max(1, w - 2))
    yv = Y[v]
    dyA = yv - Y[v - 1]
    dyB = Y[np.minimum(v + 1, w - 1)] - yv
    LA = np.sqrt(1.0 + dyA * dyA)
    LB = np.sqrt(1.0 + dyB * dyB)
    cr = dyB - dyA
    ok = ok & (cr != 0.0)
    sg = np.where(cr > 0.0, -1.0, 1.0)
    u1 = (sg * (-dyA)) / LA
    w1 = sg / LA
    u2 = (sg * (-dyB)) / LB
    w2 = sg / LB
    uc = (px + 0.5) - v
    ymin = np.full(uc.shape, _BIG)
    ymax = np.full(uc.shape, -_BIG)
    hit = np.zeros(uc.shape, dtype=bool)
    for (ua, ya, ub, yb) in ((0.0, 0.0, u1, w1), (0.0, 0.0, u2, w2), (u1, w1, u2, w2)):
        ua = np.broadcast_to(np.asarray(ua, dtype=np.float64), uc.shape)
        ya = np.broadcast_to(np.asarray(ya, dtype=np.float64), uc.shape)
        lo_u = np.minimum(ua, ub)
        hi_u = np.maximum(ua, ub)
        cross = (ua != ub) & (lo_u <= uc) & (uc <= hi_u)
        with np.errstate(divide='ignore', invalid='ignore'):
            yc = ya + ((uc - ua) * (yb - ya)) / (ub - ua)
        ymin = np.where(cross & (yc < ymin), yc, ymin)
        ymax = np.where(cross & (yc > ymax), yc, ymax)
        hit = hit | cross
    ok = ok & hit
    return yv + ymin, yv + ymax, ok


def spans_waveform_linear(samples, energy, w, h):
    """Column spans of the width-2 SquareCap/BevelJoin polyline through (x, y_x), x = 0..w-1.

    Slots: [segment px-1, segment px, segment px+1, bevel at vertex px, bevel at vertex px+1].
    """
    Y = drawlist_waveform_linear(samples, energy, w, h)
    return spans_polyline(Y, w, h)


def spans_polyline(Y, w, h):
    Y = np.asarray(Y, dtype=np.float64)
    px = np.arange(w, dtype=np.int64)
    spans = np.zeros((w, K_WAVEFORM, 2), dtype=np.int32)
    if w < 2:
        return spans
    for slot, seg in enumerate((px - 1, px, px + 1)):
        ylo, yhi, ok = _segment_interval(Y, seg, px, w)
        r0, r1 = _rows_from_interval(ylo, yhi, ok, h)
        spans[:, slot, 0] = r0
        spans[:, slot, 1] = r1
    for slot, vtx in enumerate((px, px + 1)):
        ylo, yhi, ok = _bevel_interval(Y, vtx, px, w)
        r0, r1 = _rows_from_interval(ylo, yhi, ok, h)
        spans[:, 3 + slot, 0] = r0
        spans[:, 3 + slot, 1] = r1
    return spans


# --------------------------------------------------------------------------- frames

def rasterize_spans(spans, pen_rgb, w, h, bg=BG):
    """Background fill + pen colour on every (column, row) inside a span -> uint8[h, w, 3]."""
    frame = np.empty((h, w, 3), dtype=np.uint8)
    frame[:] = np.asarray(bg, dtype=np.uint8)
    pen = np.asarray(pen_rgb, dtype=np.uint8)
    cols = np.nonzero((spans[:, :, 1] > spans[:, :, 0]).any(axis=1))[0]
    for px in cols:
        for k in range(spans.shape[1]):
            r0, r1 = int(spans[px, k, 0]), int(spans[px, k, 1])
            if r1 > r0:
                frame[r0:r1, px] = pen
    return frame


def mode_kind(mode: str) -> str:
    """Dispatch of rt_widget.py:840-849."""
    if mode.startswith("Waveform"):
        return "waveform_circular" if "Circular" in mode else "waveform_linear"
    if mode.startswith("Spectrum"):
        return "spectrum_radial" if "Radial" in mode else "spectrum_linear"
    return "particles"


def frame_spans(mode, samples, spectrum, energy, w, h):
    kind = mode_kind(mode)
    if kind == "spectrum_linear":
        return spans_spectrum_linear(spectrum, energy, w, h)
    if kind == "waveform_linear":
        return spans_waveform_linear(samples, energy, w, h)
    if kind == "particles":
        return spans_particles(spectrum, w, h)
    raise NotImplementedError(f"oracle rasteriser does not cover mode {mode!r}")


def render_frame(mode, samples, spectrum, energy, pen_rgb, w, h, bg=BG):
    """One exported frame (rt_widget.py:821-852) -> uint8[h, w, 3]; ``bg``: (13, 13, 18) or the composed background
    image uint8 [h, w, 3] (compose_background)."""
    return rasterize_spans(frame_spans(mode, samples, spectrum, energy, w, h), pen_rgb, w, h, bg)


def psnr(a: np.ndarray, b: np.ndarray) -> float:
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return float('inf') if mse == 0.0 else 10.0 * np.log10(255.0 * 255.0 / mse)


# --------------------------------------------------------------------------- Spectrum - Radial (closed stroke + disc)
#
# Same coverage rule as above (pixel centre inside the closed outline), evaluated per primitive and per pixel row:
#   * closed polyline stroke = one rectangle per segment (half-width 1, no caps: the path is closed) + one bevel
#     triangle per vertex on the outer side of the turn; each primitive is a convex polygon whose intersection with
#     the row line y = r + 0.5 is an x interval [xmin, xmax]; pixels ceil(xmin - .5) .. floor(xmax - .5) are lit;
#   * the centre disc (rt_widget.py:1056-1096, QColor(30, 30, 40) or the centre image) is filled through an antialiased
#     clip: coverage_aa / draw_center_disc at the end of this file (Qt coverage is parity-unpinned);
#   * feather rings (rt_widget.py:1098-1117, only with feather_enabled, which has no UI control): six aliased even-odd
#     fills of QColor(13, 13, 18, alpha) blended source-over in ring order with Qt's premultiplied ARGB32 arithmetic,
#     d' = div255(c a) + div255(d (255 - a)), div255(x) = (x + (x >> 8) + 0x80) >> 8 (parity-unpinned like the rest).
# Masks are bit planes uint32[H, ceil(W / 32)], bit (x & 31) of word x >> 5.  All geometry is float64 with the
# operation order below; the HIP kernels (csrc/radial.hip) use the same order with contraction disabled.

DISC_RGB = (30, 30, 40)     # rt_widget.py:1092/1094


def _convex_row_interval(px, py, yc):
    """x range of the intersection of a convex polygon (vertex lists px, py) with the line y = yc, or None."""
    n = len(px)
    xmin, xmax, hit = _BIG, -_BIG, False
    for e in range(n):
        x0, y0, x1, y1 = px[e], py[e], px[(e + 1) % n], py[(e + 1) % n]
        if y0 == y1:
            if y0 == yc:
                xmin = min(xmin, x0, x1)
                xmax = max(xmax, x0, x1)
                hit = True
        elif (yc - y0) * (yc - y1) <= 0.0:
            x = x0 + ((yc - y0) * (x1 - x0)) / (y1 - y0)
            xmin = min(xmin, x)
            xmax = max(xmax, x)
            hit = True
    return (xmin, xmax) if hit else None


def _or_span(mask, r, c0, c1, w):
    c0 = max(0, int(c0))
    c1 = min(w, int(c1))
    for c in range(c0, c1):
        mask[r, c >> 5] |= np.uint32(1 << (c & 31))


def _raster_convex(mask, px, py, w, h):
    ymin, ymax = min(py), max(py)
    r0 = max(0, int(math.ceil(ymin - 0.5)))
    r1 = min(h, int(math.floor(ymax - 0.5)) + 1)
    for r in range(r0, r1):
        iv = _convex_row_interval(px, py, r + 0.5)
        if iv is None:
            continue
        xmin, xmax = iv
        xmin = min(max(xmin, -_BIG), _BIG)
        xmax = min(max(xmax, -_BIG), _BIG)
        _or_span(mask, r, math.ceil(xmin - 0.5), math.floor(xmax - 0.5) + 1, w)


def stroke_mask_closed(verts, w, h, half_width=1.0, dx=0.0, dy=0.0, closed=True, round_join=False):
    """Bit plane of a BevelJoin stroke (pen width 2 * half_width) of the polygon through ``verts`` float64[n, 2]
    translated by (dx, dy); closed=False: open polyline with SquareCap ends (`_draw_shadow_path` of the envelope);
    round_join=True: RoundCap / RoundJoin pen = plain segment rectangles + a disc of radius half_width per vertex."""
    import math as _m
    V = np.asarray(verts, dtype=np.float64)
    n = V.shape[0]
    mask = np.zeros((h, (w + 31) // 32), dtype=np.uint32)
    ux = np.zeros(n)
    uy = np.zeros(n)
    ok = np.zeros(n, dtype=bool)
    hw, tx, ty = float(half_width), float(dx), float(dy)
    for s in range(n):                       # segment s: V[s] -> V[(s + 1) % n]
        ex, ey = V[(s + 1) % n][0] - V[s][0], V[(s + 1) % n][1] - V[s][1]
        ln = _m.sqrt(ex * ex + ey * ey)
        if ln == 0.0:
            continue
        ok[s] = True
        ux[s], uy[s] = ex / ln, ey / ln
        if not closed and s == n - 1:
            continue
        ax, ay = V[s][0] + tx, V[s][1] + ty
        bx, by = V[(s + 1) % n][0] + tx, V[(s + 1) % n][1] + ty
        if not closed and not round_join and s == 0:     # SquareCap: the end segments are extended by the half width
            ax, ay = ax - ux[s] * hw, ay - uy[s] * hw
        if not closed and not round_join and s == n - 2:
            bx, by = bx + ux[s] * hw, by + uy[s] * hw
        nx, ny = -uy[s] * hw, ux[s] * hw
        _raster_convex(mask, [ax + nx, bx + nx, bx - nx, ax - nx], [ay + ny, by + ny, by - ny, ay - ny], w, h)
    for v in range(n):                       # bevel at V[v] between segments v-1 and v (or a disc: round joins and caps)
        if round_join:
            vx, vy = V[v][0] + tx, V[v][1] + ty
            yl = min(max(vy - hw, -_BIG), _BIG)
            yh = min(max(vy + hw, -_BIG), _BIG)
            for r in range(max(0, int(math.ceil(yl - 0.5))), min(h, int(math.floor(yh - 0.5)) + 1)):
                dyc = (r + 0.5) - vy
                c2 = hw * hw - dyc * dyc
                if c2 < 0.0:
                    continue
                c = _m.sqrt(c2)
                xl = min(max(vx - c, -_BIG), _BIG)
                xh = min(max(vx + c, -_BIG), _BIG)
                _or_span(mask, r, math.ceil(xl - 0.5), math.floor(xh - 0.5) + 1, w)
            continue
        if not closed and (v == 0 or v == n - 1):
            continue
        a, b = (v - 1) % n, v
        if not (ok[a] and ok[b]):
            continue
        cr = ux[a] * uy[b] - uy[a] * ux[b]
        if cr == 0.0:
            continue
        sg = -1.0 if cr > 0.0 else 1.0
        vx, vy = V[v][0] + tx, V[v][1] + ty
        _raster_convex(mask, [vx, vx + sg * (-uy[a]) * hw, vx + sg * (-uy[b]) * hw],
                       [vy, vy + sg * ux[a] * hw, vy + sg * ux[b] * hw], w, h)
    return mask


def polygon_mask(verts, w, h, max_cross=None):
    """Bit plane of the even-odd fill of the (implicitly closed) polygon ``verts``, sampled at pixel centres."""
    V = np.asarray(verts, dtype=np.float64)
    n = V.shape[0]
    mask = np.zeros((h, (w + 31) // 32), dtype=np.uint32)
    for r in range(h):
        yc = r + 0.5
        xs = []
        for e in range(n):
            x0, y0 = V[e]
            x1, y1 = V[(e + 1) % n]
            if (y0 <= yc) != (y1 <= yc):
                xs.append(x0 + ((yc - y0) * (x1 - x0)) / (y1 - y0))
        xs.sort()       # any number of crossings per row (an odd count, which only NaN vertices produce, drops the last)
        for i in range(0, len(xs) - 1, 2):
            a = min(max(xs[i], -_BIG), _BIG)
            b = min(max(xs[i + 1], -_BIG), _BIG)
            _or_span(mask, r, math.ceil(a - 0.5), math.ceil(b - 0.5), w)
    return mask


def unpack_mask(mask, w):
    bits = ((mask[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    return bits.reshape(mask.shape[0], -1)[:, :w]


def _div255(x):
    return (x + (x >> 8) + 0x80) >> 8


def blend_solid(frame, mask_bool, rgb, alpha):
    """Source-over of the constant colour (rgb, alpha) on the pixels of ``mask_bool``, in place (Qt BYTE_MUL form)."""
    d = frame[mask_bool].astype(np.uint32)
    c = np.asarray(rgb, dtype=np.uint32)
    a = np.uint32(alpha)
    frame[mask_bool] = (_div255(c * a) + _div255(d * (np.uint32(255) - a))).astype(np.uint8)


def fill_colour_over_background(rgba, blend="normal", bg=BG):
    """Result of `p.fillPath(path, fill_color)` on the plain background for the three blends of `_apply_fill`
    (rt_widget.py:856-864): Qt premultiplied ARGB32 SourceOver / Plus / Multiply over an opaque destination."""
    r, g, b, a = (int(v) for v in rgba)
    out = []
    for c, d in zip((r, g, b), bg):
        s = int(_div255(c * a))
        if blend == "add":
            out.append(min(255, s + d))
        elif blend == "multiply":
            out.append(int(_div255(s * d + d * (255 - a))))
        else:
            out.append(s + int(_div255(d * (255 - a))))
    return tuple(out)


def render_radial_frame(path_verts, env_verts, pen_rgb, w, h, bg=BG, rings=None, fill_rgb=None, image_pm=None,
                        placement=None, base=None):
    """One "Spectrum - Radial" frame: background (or the composed background image ``base``), radial stroke in the pen
    colour, the antialiased centre disc on top (plain fill, or the centre image ``image_pm`` at ``placement``), then
    the feather rings ``[(vertices, alpha)]`` (radial_ref.feather_ring_paths) blended in order."""
    frame = np.empty((h, w, 3), dtype=np.uint8)
    frame[:] = np.asarray(bg, dtype=np.uint8) if base is None else base
    if fill_rgb is not None:                 # `_apply_fill` comes first (rt_widget.py:977)
        frame[unpack_mask(polygon_mask(path_verts, w, h), w)] = np.asarray(fill_rgb, dtype=np.uint8)
    frame[unpack_mask(stroke_mask_closed(path_verts, w, h), w)] = np.asarray(pen_rgb, dtype=np.uint8)
    if env_verts is not None:
        draw_center_disc(frame, env_verts, w, h, image_pm, placement)
    for verts, alpha in (rings or []):
        blend_solid(frame, unpack_mask(polygon_mask(verts, w, h), w), bg, alpha)
    return frame


def render_wave_circular_frame(path_verts, pen_rgb, w, h, bg=BG):
    """One "Waveform - Circular" frame (rt_widget.py:866-902): background (colour or composed image) and the stroked
    closed path."""
    frame = np.empty((h, w, 3), dtype=np.uint8)
    frame[:] = np.asarray(bg, dtype=np.uint8)
    frame[unpack_mask(stroke_mask_closed(path_verts, w, h), w)] = np.asarray(pen_rgb, dtype=np.uint8)
    return frame


# ----------------------------------------------------------------------------- glow decoration
# `_draw_glow_path` -> `_draw_glow_offscreen` (widgets/rt_widget.py:598-616, :634-653; the export always runs with
# _offscreen_paint, export/exporter.py:118): after fill and path shadow, before the path's own stroke, the closed path is
# stroked once per layer with an antialiased RoundCap / RoundJoin pen, CompositionMode_Plus (round_stroke_coverage below;
# Qt coverage is parity-unpinned).
# Pen colours, widths, composition mode and call order are pinned by tests/golden/glow_*.npz.

def qcolor_alpha_f(alpha: int) -> float:
    """QColor.alphaF() of an 8-bit alpha: Qt 6 keeps a * 0x101 and returns it / 65535 as a C float (the stand-in QColor
    of tests/golden/make_golden.py models the same; PySide6 is not installed, so this step is unpinned against Qt)."""
    return float(np.float32(int(alpha) * 257) / np.float32(65535))


def glow_pens(energy: float, rgba=(0xA0, 0xFF, 0xFF, 0x80), radius: int = 22, strength: float = 0.8, layers: int = 8):
    """[(pen width, alpha)] of the layers `_draw_glow_offscreen` paints in one frame, in paint order (rt_widget.py:599-612,
    :635-648); empty when the glow does not paint at all."""
    if radius <= 0 or strength <= 0.0:
        return []
    audio_intensity = 0.15 + 0.85 * max(0.0, min(1.0, energy))
    intensity = strength * audio_intensity
    if intensity <= 0.0:
        return []
    base_alpha = max(0.0, min(1.0, qcolor_alpha_f(rgba[3])))
    n = max(4, min(16, layers))
    out = []
    for i in range(n):
        t = i / float(max(1, n - 1))
        falloff = (1.0 - t) ** 2
        a = int(255 * base_alpha * intensity * 0.55 * falloff)
        if a <= 0:
            continue
        out.append((max(3, int(2 + radius * (0.35 + 0.9 * t))), a))
    return out


def add_solid(frame, mask_bool, rgb, alpha):
    """CompositionMode_Plus of the constant colour (rgb, alpha) on the pixels of ``mask_bool``, in place: the
    premultiplied source div255(c a) is added per channel and saturates at 255 (Qt comp_func_Plus, opaque destination)."""
    c = np.asarray(rgb, dtype=np.uint32)
    pm = _div255(c * np.uint32(alpha))
    d = frame[mask_bool].astype(np.uint32)
    frame[mask_bool] = np.minimum(d + pm, 255).astype(np.uint8)


# Antialiased round stroke (`setRenderHint(Antialiasing)`, RoundCap / RoundJoin pen: rt_widget.py:637, :649-650).  The
# stroked shape is every point within half the pen width of the path; Qt fills it through its gray raster.  Qt's stroker
# and rasteriser are not in the reference tree (**parity unpinned**); the rule here, mirrored bit for bit by
# csrc/glow.hip, is the area a pixel shares with that shape when its rim is taken as straight across the pixel:
#   d  = distance from the pixel centre to the nearest point q of the path, n = (centre - q) / d,
#   s  = half_width - d  (signed depth of the centre inside the shape),
#   coverage = area of the unit square on the inner side of the line at depth s with normal n, in 1/256 steps (255 max).
# All float64, operation order as written (the kernels are built with contraction off).

GLOW_REACH = 0.7072          # beyond this depth a pixel is entirely inside / outside (half diagonal of a pixel)


def path_distance_field(verts, w, h, reach):
    """(d2, qx, qy) float64 [h, w]: squared distance of every pixel centre to the nearest point of the closed polyline
    ``verts`` and that point; exact wherever the distance is at most ``reach`` (inf / 0 where nothing is that near).
    Ties go to the segment with the lowest index."""
    V = np.asarray(verts, dtype=np.float64)
    n = V.shape[0]
    d2 = np.full((h, w), np.inf)
    qx = np.zeros((h, w))
    qy = np.zeros((h, w))
    for i in range(n):
        ax, ay = float(V[i, 0]), float(V[i, 1])
        bx, by = float(V[(i + 1) % n, 0]), float(V[(i + 1) % n, 1])
        if not (ax == ax and ay == ay and bx == bx and by == by):
            continue
        ex, ey = bx - ax, by - ay
        l2 = ex * ex + ey * ey
        inv = 1.0 / l2 if l2 > 0.0 else 0.0
        x0 = max(0, int(math.floor(min(ax, bx) - reach - 1.0)))
        x1 = min(w, int(math.ceil(max(ax, bx) + reach + 1.0)) + 1)
        y0 = max(0, int(math.floor(min(ay, by) - reach - 1.0)))
        y1 = min(h, int(math.ceil(max(ay, by) + reach + 1.0)) + 1)
        if x1 <= x0 or y1 <= y0:
            continue
        px = np.arange(x0, x1, dtype=np.float64)[None, :] + 0.5
        py = np.arange(y0, y1, dtype=np.float64)[:, None] + 0.5
        t = ((px - ax) * ex + (py - ay) * ey) * inv
        cx = np.where(t <= 0.0, ax, np.where(t >= 1.0, ax + ex, ax + t * ex))
        cy = np.where(t <= 0.0, ay, np.where(t >= 1.0, ay + ey, ay + t * ey))
        dd = (px - cx) * (px - cx) + (py - cy) * (py - cy)
        sl = (slice(y0, y1), slice(x0, x1))
        better = dd < d2[sl]
        d2[sl] = np.where(better, dd, d2[sl])
        qx[sl] = np.where(better, cx, qx[sl])
        qy[sl] = np.where(better, cy, qy[sl])
    return d2, qx, qy


def round_stroke_coverage(d2, qx, qy, half_width):
    """uint8 [h, w] coverage of the round stroke of half width ``half_width`` from a distance field."""
    h, w = d2.shape
    px = np.arange(w, dtype=np.float64)[None, :] + 0.5
    py = np.arange(h, dtype=np.float64)[:, None] + 0.5
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.sqrt(d2)
        s = half_width - d
        inv_d = 1.0 / d                     # one reciprocal per pixel, shared by both components and by every layer
        nx = np.abs((px - qx) * inv_d)
        ny = np.abs((py - qy) * inv_d)
        a = np.maximum(nx, ny)
        b = np.minimum(nx, ny)
        inv_a = 1.0 / a
        inv_2ab = 1.0 / ((2.0 * a) * b)
        u = s + (a + b) / 2.0
        f_lo = (u * u) * inv_2ab
        f_mid = (u - b / 2.0) * inv_a
        r = (a + b) - u
        f_hi = 1.0 - (r * r) * inv_2ab
        f = np.where(u <= 0.0, 0.0, np.where(u >= a + b, 1.0, np.where(u < b, f_lo, np.where(u <= a, f_mid, f_hi))))
        c = np.floor(f * 256.0)
    c = np.where(np.isfinite(c), c, 0.0)
    c = np.minimum(np.maximum(c, 0.0), 255.0)
    c = np.where(s >= GLOW_REACH, 255.0, np.where(s <= -GLOW_REACH, 0.0, c))
    c = np.where(np.isfinite(d2), c, 0.0)
    return c.astype(np.uint8)


def plus_cov(frame, pm_rgb, cov):
    """CompositionMode_Plus of the premultiplied colour with a per-pixel coverage, in place (Qt comp_func_solid_Plus with
    const_alpha = coverage): r = min(255, d + s); d' = INTERPOLATE_PIXEL_255(r, cov, d, 255 - cov)."""
    d = frame.astype(np.uint32)
    c = cov.astype(np.uint32)[..., None]
    r = np.minimum(d + np.asarray(pm_rgb, dtype=np.uint32), 255)
    v = r * c + d * (255 - c)
    out = (v + (v >> 8) + 0x80) >> 8
    m = cov > 0
    frame[m] = out[m].astype(np.uint8)
    return frame


def apply_glow(frame, path_verts, energy, keep_bool, w, h, rgba=(0xA0, 0xFF, 0xFF, 0x80), radius=22, strength=0.8,
               layers=8):
    """Glow of the closed path on ``frame`` (in place), restricted to the pixels ``keep_bool`` (those that neither the
    stroke nor the fully covered part of the disc, painted afterwards with opaque colours, cover): one antialiased
    RoundCap / RoundJoin stroke per layer, CompositionMode_Plus, in the reference's layer order."""
    pens = glow_pens(energy, rgba, radius, strength, layers)
    if not pens:
        return frame
    reach = max(wd for wd, _ in pens) / 2.0 + 1.0
    d2, qx, qy = path_distance_field(path_verts, w, h, reach)
    for width, alpha in pens:
        cov = round_stroke_coverage(d2, qx, qy, width / 2.0)
        cov = np.where(keep_bool, cov, 0).astype(np.uint8)
        pm = _div255(np.asarray(rgba[:3], dtype=np.uint32) * np.uint32(alpha))
        plus_cov(frame, pm, cov)
    return frame


# ----------------------------------------------------------------------------- antialiased polygon coverage
# `_draw_center_image_and_mask` turns QPainter.Antialiasing on before `setClipPath(path)` / `fillPath(path, ...)`
# (widgets/rt_widget.py:1056-1058, :1092-1094): Qt's raster engine then rasterises the path with its gray-raster
# (qgrayraster.c, the FreeType "smooth" rasteriser): vertices are rounded to 26.6 fixed point, scaled to 1/256 pixel, and
# every pixel gets the exact signed area the outline covers of it, in 1/512 steps, folded by the fill rule and clamped to
# 255.  Qt's source is not in the reference tree, so this layer is **parity unpinned**; the rule implemented here (and,
# bit for bit, by csrc/image.hip) is that algorithm with one canonical rounding: where an edge crosses a pixel-row or
# pixel-column boundary the crossing point is rounded DOWN to the 1/256 grid from the edge's own end points (FreeType
# distributes the same remainders incrementally).  All arithmetic is integer, so the result does not depend on the order
# in which edges are accumulated.

def _fixed_26_6(v: float) -> int:
    """qreal_to_fixed_26_6 = qRound(v * 64), then scaled to the gray raster's 1/256 pixel units."""
    return int(math.floor(v * 64.0 + 0.5)) * 4


def _cells_of_edge(x0, y0, x1, y1):
    """(ex, ey, dcover, darea) of every pixel cell the edge (integers, 1/256 px) passes through."""
    if y0 == y1:
        return
    sign = 1 if y1 > y0 else -1
    if sign < 0:
        x0, y0, x1, y1 = x1, y1, x0, y0            # canonical: upwards in y; the sign carries the direction
    dx, dy = x1 - x0, y1 - y0                      # dy > 0
    if x0 <= x1:
        xl, yl_x, xr, yr_x = x0, y0, x1, y1        # canonical by x for the column crossings
    else:
        xl, yl_x, xr, yr_x = x1, y1, x0, y0
    for ey in range(y0 >> 8, ((y1 - 1) >> 8) + 1):
        ya = max(y0, ey << 8)
        yb = min(y1, (ey + 1) << 8)
        xa = x0 if ya == y0 else x0 + ((ya - y0) * dx) // dy
        xb = x1 if yb == y1 else x0 + ((yb - y0) * dx) // dy
        if xa == xb:
            ex = xa >> 8
            yield ex, ey, sign * (yb - ya), sign * (yb - ya) * 2 * (xa - (ex << 8))
            continue
        lo, hi = (xa, xb) if xa < xb else (xb, xa)
        ylo, yhi = (ya, yb) if xa < xb else (yb, ya)   # y at the piece's left / right end
        for ex in range(lo >> 8, ((hi - 1) >> 8) + 1):
            cxa = max(lo, ex << 8)
            cxb = min(hi, (ex + 1) << 8)

            def y_at(x, x_end_lo=lo, x_end_hi=hi):
                if x == x_end_lo:
                    return ylo
                if x == x_end_hi:
                    return yhi
                y = yl_x + ((x - xl) * (yr_x - yl_x)) // (xr - xl)
                return min(max(y, ya), yb)
            pya, pyb = y_at(cxa), y_at(cxb)
            d = (pyb - pya) if xa < xb else (pya - pyb)        # upwards along the canonical direction
            yield ex, ey, sign * d, sign * d * ((cxa - (ex << 8)) + (cxb - (ex << 8)))


def coverage_aa(verts, w, h):
    """uint8 [h, w]: antialiased even-odd coverage of the (implicitly closed) polygon ``verts`` float64 [n, 2]."""
    V = np.asarray(verts, dtype=np.float64)
    n = V.shape[0]
    X = [_fixed_26_6(float(v)) for v in V[:, 0]]
    Y = [_fixed_26_6(float(v)) for v in V[:, 1]]
    cover = np.zeros((h, w + 1), dtype=np.int64)          # column 0: everything left of the frame
    area = np.zeros((h, w), dtype=np.int64)
    for e in range(n):
        e1 = (e + 1) % n
        for ex, ey, dc, da in _cells_of_edge(X[e], Y[e], X[e1], Y[e1]):
            if ey < 0 or ey >= h or ex >= w:
                continue
            if ex < 0:
                cover[ey, 0] += dc
            else:
                cover[ey, ex + 1] += dc
                area[ey, ex] += da
    acc = np.cumsum(cover, axis=1)[:, 1:]
    val = np.abs(acc * 512 - area) >> 9
    val &= 511
    val = np.where(val > 256, 512 - val, val)
    return np.minimum(val, 255).astype(np.uint8)


def byte_mul(x, a):
    """Qt's BYTE_MUL on one channel: x * a / 255 rounded the way qdrawhelper does."""
    t = x * a
    return (t + (t >> 8) + 0x80) >> 8


def blend_cov(frame, src_rgba_pm, cov):
    """comp_func_SourceOver with a per-pixel coverage (Qt raster engine, spans of an antialiased clip / fill):
    s = BYTE_MUL(src, cov) on the premultiplied source, d' = s + BYTE_MUL(d, 255 - alpha(s)).  ``src_rgba_pm``
    uint8 [h, w, 4] (or one (r, g, b, a)) premultiplied, ``cov`` uint8 [h, w]; in place on ``frame`` uint8 [h, w, 3]."""
    c = cov.astype(np.uint32)[..., None]
    src = np.broadcast_to(np.asarray(src_rgba_pm, dtype=np.uint32), frame.shape[:2] + (4,))
    s = byte_mul(src, c)
    d = frame.astype(np.uint32)
    out = s[..., :3] + byte_mul(d, 255 - s[..., 3:4])
    m = cov > 0
    frame[m] = out[m].astype(np.uint8)
    return frame


def disc_fill_coverage(cov):
    """`fillPath(path, colour)` under `setClipPath(path)` with antialiasing: the fill's coverage is intersected with the
    clip's (qt_intersect_spans: qt_div_255(a * b)); both come from the same path."""
    c = cov.astype(np.uint32)
    return byte_mul(c, c).astype(np.uint8)


# ----------------------------------------------------------------------------- QImage.scaled(..., SmoothTransformation)
# `img.scaled(out_w, out_h, Qt.IgnoreAspectRatio, Qt.SmoothTransformation)` (widgets/rt_widget.py:1090, :567) ends in
# qSmoothScaleImage (Qt 6 src/gui/painting/qimagescale.cpp, the Imlib2 scaler) on a premultiplied ARGB32 image.  Restated
# from that algorithm's published form (generic C path); Qt's source is not in the reference tree: **parity unpinned**.
# Up-scaling an axis: 8-bit bilinear weights from 16.16 positions; down-scaling: area average with 14-bit weights.

def _scale_points(s, d):
    """(pos[d], ap[d], up) of one axis: source index and weight word per destination index."""
    up = d >= s
    inc = (s << 16) // d
    pos = np.empty(d, dtype=np.int64)
    ap = np.empty(d, dtype=np.int64)
    if up:
        val = (0x8000 * s) // d - 0x8000
        for i in range(d):
            p = val >> 16
            pos[i] = max(0, p)
            ap[i] = 0 if (p < 0 or p >= s - 1) else ((val >> 8) & 0xff)
            val += inc
    else:
        cp = ((d << 14) + s - 1) // s
        val = 0
        for i in range(d):
            pos[i] = val >> 16
            ap[i] = (((0x10000 - (val & 0xffff)) * cp) >> 16) | (cp << 16)
            val += inc
    return pos, ap, up


def _interp256(x, a, y, b):
    return (x * a + y * b) >> 8          # INTERPOLATE_PIXEL_256 per channel


def _area_axis(fetch, start, ap_word, limit):
    """qt_qimageScaleAARGBA_helper along one axis: fetch(i) -> int64 [4]; returns the 14-bit weighted sum."""
    xyap, c = int(ap_word) & 0xffff, int(ap_word) >> 16
    i = start
    acc = fetch(min(i, limit)) * xyap
    j = (1 << 14) - xyap
    while j > c:
        i += 1
        acc = acc + fetch(min(i, limit)) * c
        j -= c
    i += 1
    acc = acc + fetch(min(i, limit)) * j
    return acc


def smooth_scale_pixel(img, dw, dh, x, y, _cache={}):
    """One pixel (x, y) of ``img`` uint8 [sh, sw, 4] (premultiplied r, g, b, a) smooth-scaled to dw x dh -> int [4]."""
    sh, sw = img.shape[:2]
    key = (sw, sh, dw, dh)
    if key not in _cache:
        _cache.clear()
        _cache[key] = (_scale_points(sw, dw), _scale_points(sh, dh))
    (xpos, xap, xup), (ypos, yap, yup) = _cache[key]
    px = lambda xx, yy: img[yy, xx].astype(np.int64)
    sx, sy = int(xpos[x]), int(ypos[y])
    if xup and yup:
        xa, ya = int(xap[x]), int(yap[y])
        if ya > 0:
            if xa > 0:
                top = _interp256(px(sx, sy), 256 - xa, px(sx + 1, sy), xa)
                bot = _interp256(px(sx, sy + 1), 256 - xa, px(sx + 1, sy + 1), xa)
                return _interp256(top, 256 - ya, bot, ya)
            return _interp256(px(sx, sy), 256 - ya, px(sx, sy + 1), ya)
        if xa > 0:
            return _interp256(px(sx, sy), 256 - xa, px(sx + 1, sy), xa)
        return px(sx, sy)
    if xup and not yup:                       # qt_qimageScaleAARGBA_up_x_down_y
        col = lambda xx: _area_axis(lambda i: px(xx, i), sy, yap[y], sh - 1)
        r = col(sx)
        xa = int(xap[x])
        if xa > 0:
            rr = col(sx + 1)
            r = (r * (256 - xa) + rr * xa) >> 8
        return r >> 14
    if not xup and yup:                       # qt_qimageScaleAARGBA_down_x_up_y
        row = lambda yy: _area_axis(lambda i: px(i, yy), sx, xap[x], sw - 1)
        r = row(sy)
        ya = int(yap[y])
        if ya > 0:
            rr = row(sy + 1)
            r = (r * (256 - ya) + rr * ya) >> 8
        return r >> 14
    # down both: rows of x-averaged values (>> 4), weighted along y
    rowsum = lambda yy: _area_axis(lambda i: px(i, yy), sx, xap[x], sw - 1) >> 4
    return _area_axis(rowsum, sy, yap[y], sh - 1) >> 24


def smooth_scale(img, dw, dh):
    """Whole image; small sizes only (pure Python)."""
    out = np.empty((dh, dw, 4), dtype=np.uint8)
    for y in range(dh):
        for x in range(dw):
            out[y, x] = smooth_scale_pixel(img, dw, dh, x, y)
    return out


def premultiply(rgba):
    """qPremultiply of uint8 [..., 4] straight-alpha (r, g, b, a)."""
    p = np.asarray(rgba, dtype=np.uint32)
    a = p[..., 3:4]
    out = p.copy()
    out[..., :3] = byte_mul(p[..., :3], a)
    return out.astype(np.uint8)


# ----------------------------------------------------------------------------- centre image placement
def center_image_placement(clip_verts, img_w, img_h, zoom=100, center_motion=0, energy=0.0, phase=0.0):
    """(out_w, out_h, dx, dy) of `p.drawImage(dx, dy, img.scaled(out_w, out_h, ...))` (rt_widget.py:1065-1090).
    path.boundingRect() of a polyline path = min / max of its vertices; QRectF.adjusted(-2, -2, 2, 2)."""
    V = np.asarray(clip_verts, dtype=np.float64)
    minx, maxx, miny, maxy = float(V[:, 0].min()), float(V[:, 0].max()), float(V[:, 1].min()), float(V[:, 1].max())
    rx, ry, rw, rh = minx + -2.0, miny + -2.0, ((maxx - minx) + 2.0) - -2.0, ((maxy - miny) + 2.0) - -2.0
    rect_w = max(1, int(rw))
    rect_h = max(1, int(rh))
    scale = max(rect_w / float(img_w), rect_h / float(img_h))
    user_zoom = max(0.1, float(zoom) / 100.0)
    scale *= user_zoom
    cm = float(center_motion) / 100.0
    if cm > 0.0:
        motion_zoom = 1.0 + 0.12 * cm * (0.4 + 0.6 * energy) * (0.5 + 0.5 * np.sin(phase * 1.7))
        scale *= max(0.1, motion_zoom)
    out_w = int(max(1, round(img_w * scale)))
    out_h = int(max(1, round(img_h * scale)))
    cx_rect = rx + rw / 2.0
    cy_rect = ry + rh / 2.0
    dx = int(round(cx_rect - out_w / 2.0))
    dy = int(round(cy_rect - out_h / 2.0))
    if cm > 0.0:
        j = int(cm * 6)
        dx += int(round(j * np.sin(phase * 2.3)))
        dy += int(round(j * np.cos(phase * 1.9)))
    return out_w, out_h, dx, dy


def draw_center_disc(frame, env_verts, w, h, image_pm=None, placement=None):
    """The centre disc on ``frame`` (in place): antialiased clip of the envelope path, then either the (30, 30, 40) fill
    (coverage of fill and clip multiplied) or the smooth-scaled centre image at ``placement`` = (out_w, out_h, dx, dy)."""
    cov = coverage_aa(env_verts, w, h)
    if image_pm is None:
        return blend_cov(frame, DISC_RGB + (255,), disc_fill_coverage(cov))
    out_w, out_h, dx, dy = placement
    ys, xs = np.nonzero(cov)
    src = np.zeros((h, w, 4), dtype=np.uint8)
    keep = np.zeros((h, w), dtype=bool)
    for y, x in zip(ys, xs):
        ix, iy = int(x) - dx, int(y) - dy
        if 0 <= ix < out_w and 0 <= iy < out_h:
            src[y, x] = smooth_scale_pixel(image_pm, out_w, out_h, ix, iy)
            keep[y, x] = True
    return blend_cov(frame, src, np.where(keep, cov, 0).astype(np.uint8))


# ----------------------------------------------------------------------------- background image
def background_plan(img_w, img_h, w, h, scale_mode="fill", offset=(0, 0), dim_percent=0):
    """(scaled_w, scaled_h, tile, x, y, dim_alpha) of `_draw_bg_image` / `_scaled_bg_for` (rt_widget.py:519-575);
    QSize::scaled in Qt's integer arithmetic."""
    mode = (scale_mode or "fill").lower()
    d = max(0, min(100, int(dim_percent)))
    dim_alpha = int(255 * (d / 100.0)) if d > 0 else 0
    if mode == "tile":
        return img_w, img_h, True, offset[0], offset[1], dim_alpha
    if mode == "center":
        sw, sh = img_w, img_h
    elif mode == "stretch":
        sw, sh = w, h
    else:
        rw = (h * img_w) // img_h
        use_height = (rw <= w) if mode == "fit" else (rw >= w)
        sw, sh = (rw, h) if use_height else (w, (w * img_h) // img_w)
    sw, sh = max(sw, 1), max(sh, 1)
    return sw, sh, False, (w - sw) // 2 + offset[0], (h - sh) // 2 + offset[1], dim_alpha


def compose_background(image_pm, w, h, scale_mode="fill", offset=(0, 0), dim_percent=0):
    """uint8 [h, w, 3]: fillRect (13, 13, 18), the background image (premultiplied uint8 [ih, iw, 4]) smooth-scaled and
    drawn source-over at its place (or tiled), then the dim rectangle (rt_widget.py:829-830, :519-545)."""
    ih, iw = image_pm.shape[:2]
    sw, sh, tile, x0, y0, dim_alpha = background_plan(iw, ih, w, h, scale_mode, offset, dim_percent)
    frame = np.empty((h, w, 3), dtype=np.uint8)
    frame[:] = np.asarray(BG, dtype=np.uint8)
    for y in range(h):
        for x in range(w):
            ix, iy = x - x0, y - y0
            if tile:
                if ix < 0 or iy < 0 or (ix // iw) * iw >= w or (iy // ih) * ih >= h:
                    continue
                s = image_pm[iy % ih, ix % iw].astype(np.int64)
            else:
                if not (0 <= ix < sw and 0 <= iy < sh):
                    continue
                s = (image_pm[iy, ix].astype(np.int64) if (sw, sh) == (iw, ih)
                     else smooth_scale_pixel(image_pm, sw, sh, ix, iy))
            d = frame[y, x].astype(np.int64)
            frame[y, x] = (s[:3] + byte_mul(d, 255 - s[3])).astype(np.uint8)
    if dim_alpha > 0:
        frame[:] = byte_mul(frame.astype(np.int64), 255 - dim_alpha).astype(np.uint8)
    return frame
