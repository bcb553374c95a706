"""CPU-only tests: host logic against the oracle, C-ABI library load/exports, raster oracle self-consistency."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

from conftest import REPO
from aurora_amd import _lib, engine, synth
from oracle import analysis_ref as A
from oracle import raster_ref as R


# ------------------------------------------------------------------ index maps must be bit-exact

@pytest.mark.parametrize("n,sr,fps", [(48000 * 240, 48000, 60), (44100 * 30, 44100, 30), (22173, 44100, 60),
                                      (500, 48000, 60), (30000, 48000, 24), (48000 * 3600, 48000, 60)])
def test_frame_plan_matches_oracle(n, sr, fps):
    hop, total, pos = engine.frame_plan(n, sr, fps)
    if total > 20000:       # python-loop oracle on a prefix + tail only
        t = np.concatenate([np.arange(5000), np.arange(total - 5000, total)])
        want = np.array([int(max(0, min(n - 1, (i / fps) * sr))) for i in t], dtype=np.int64)
        assert np.array_equal(pos[t], want)
        assert hop == max(1, int(sr / fps)) and total == int(np.ceil(float(n / sr) * fps))
    else:
        hop_o, total_o, pos_o = A.frame_plan(n, sr, fps)
        assert (hop, total) == (hop_o, total_o)
        assert np.array_equal(pos, pos_o)


@pytest.mark.parametrize("w", [160, 1000, 1280, 1920, 3840])
def test_col2samp_matches_oracle(w):
    assert np.array_equal(engine.waveform_col2samp(w), R.waveform_col2samp(w))


def test_mode_dispatch():
    for m in ("Spectrum - Linear", "Spectrum - Radial", "Waveform - Linear", "Waveform - Circular", "Particles", ""):
        assert engine.mode_kind(m) == R.mode_kind(m)


# ------------------------------------------------------------------ C ABI library

def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "aurora_hip.h")).read()
    declared = set(re.findall(r"\b(aur_[a-z_0-9]+)\s*\(", header))
    declared -= {"aur_ctx", "aur_scan_params"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.aur_version() == _lib.ABI_VERSION
    assert isinstance(lib.aur_last_error(), bytes)


def test_scan_params_layout_matches_header():
    # struct aur_scan_params: 4 doubles, int, 2x uint8[4], (pad), 4 doubles
    assert ctypes.sizeof(_lib.ScanParams) == 80
    assert _lib.ScanParams.grad_curve.offset == 32
    assert _lib.ScanParams.grad_a.offset == 36 and _lib.ScanParams.grad_b.offset == 40
    assert _lib.ScanParams.flux_alpha.offset == 48 and _lib.ScanParams.phase0.offset == 72


def test_error_paths_without_gpu():
    lib = _lib.load()
    assert lib.aur_device_count(None) == -1
    assert b"null" in lib.aur_last_error()
    assert lib.aur_stft_frames(None, None, 0, 2, None, 0, 1, 800, None, None, None, None, None, None) != 0


# ------------------------------------------------------------------ raster oracle: spans == brute-force coverage

def _brute_polyline(Y, w, h):
    """Pixel-centre-in-stroke-outline test written directly from the coverage rule (independent of the
    interval algebra in spans_polyline): rectangles with square caps + bevel triangles."""
    img = np.zeros((h, w), dtype=bool)
    py, px = np.mgrid[0:h, 0:w]
    cx, cy = px + 0.5, py + 0.5
    P = [(float(x), float(Y[x])) for x in range(w)]
    for s in range(w - 1):
        (x0, y0), (x1, y1) = P[s], P[s + 1]
        dx, dy = x1 - x0, y1 - y0
        L = np.hypot(dx, dy)
        ux, uy = dx / L, dy / L
        rx, ry = cx - x0, cy - y0
        along = rx * ux + ry * uy
        across = -rx * uy + ry * ux
        lo = -1.0 if s == 0 else 0.0
        hi = L + (1.0 if s == w - 2 else 0.0)
        img |= (along >= lo) & (along <= hi) & (np.abs(across) <= 1.0)
    for v in range(1, w - 1):
        dA = np.array([1.0, Y[v] - Y[v - 1]])
        dB = np.array([1.0, Y[v + 1] - Y[v]])
        cr = dA[0] * dB[1] - dA[1] * dB[0]
        if cr == 0:
            continue
        sg = -1.0 if cr > 0 else 1.0
        nA = sg * np.array([-dA[1], dA[0]]) / np.hypot(*dA)
        nB = sg * np.array([-dB[1], dB[0]]) / np.hypot(*dB)
        p0 = np.array([float(v), float(Y[v])])
        p1, p2 = p0 + nA, p0 + nB

        def side(a, b):
            return (b[0] - a[0]) * (cy - a[1]) - (b[1] - a[1]) * (cx - a[0])

        d0, d1, d2 = side(p0, p1), side(p1, p2), side(p2, p0)
        img |= ((d0 >= 0) & (d1 >= 0) & (d2 >= 0)) | ((d0 <= 0) & (d1 <= 0) & (d2 <= 0))
    return img


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_polyline_spans_equal_brute_force(seed):
    rng = np.random.default_rng(seed)
    w, h = 200, 120
    # irrational-ish coordinates so no pixel centre sits on an outline edge (ties are measure zero)
    Y = (h / 2 + np.cumsum(rng.standard_normal(w) * [0.3, 3.0, 12.0][seed]) + 0.1234567).astype(np.float32).astype(np.float64)
    Y = np.clip(Y, -20, h + 20)
    spans = R.spans_polyline(Y, w, h)
    got = np.zeros((h, w), dtype=bool)
    for x in range(w):
        for k in range(spans.shape[1]):
            got[spans[x, k, 0]:spans[x, k, 1], x] = True
    want = _brute_polyline(Y, w, h)
    assert np.array_equal(got, want), int(np.sum(got != want))


def test_flat_waveform_and_bar_rules():
    w, h = 320, 200
    sp = R.spans_polyline(np.full(w, h / 2.0), w, h)
    f = R.rasterize_spans(sp, (255, 0, 0), w, h)
    lit = (f != np.array(R.BG, np.uint8)).any(axis=2)
    assert lit[h // 2 - 1].all() and lit[h // 2].all() and lit.sum() == 2 * w     # silence: 2-px line, full width
    spec = np.zeros(1025, np.float32)
    spec[3] = 2.0
    s2 = R.spans_spectrum_linear(spec, 1.0, w, h)
    f2 = R.rasterize_spans(s2, (1, 2, 3), w, h)
    lit2 = (f2 != np.array(R.BG, np.uint8)).any(axis=2)
    bw = w // 80
    assert lit2[:, 3 * bw].sum() == (h - 10) + 2 and lit2[:, 3 * bw - 1].sum() == (h - 10) + 2   # full bar + caps
    assert lit2[:, 0].sum() == 2 and lit2[:, 1].sum() == 0                                       # bh = 0 -> 2x2 dot, half clipped
    assert lit2[h - 5, 3 * bw] and not lit2[h - 4, 3 * bw]


# ------------------------------------------------------------------ worker protocol (setup failures need no GPU)

def test_worker_setup_errors(capsys, tmp_path):
    from aurora_amd.export import worker
    assert worker.main(["worker.py"]) == 2
    assert json.loads(capsys.readouterr().out.strip())["message"] == "Missing config path"
    assert worker.main(["worker.py", str(tmp_path / "nope.json")]) == 2
    assert json.loads(capsys.readouterr().out.strip())["type"] == "error"
    bad = tmp_path / "bad.json"
    bad.write_text("{not json")
    assert worker.main(["worker.py", str(bad)]) == 2
    assert "Invalid config JSON" in json.loads(capsys.readouterr().out.strip())["message"]


def test_worker_maps_config_keys_and_failures(capsys, tmp_path, monkeypatch):
    """Config keys -> Exporter arguments; a missing required key or a failing export is exit 1 with an
    error message, progress lines are integers."""
    from aurora_amd.export import exporter, worker
    cfg = {"audio_path": "a.wav", "out_path": "o.mp4", "width": "1280", "height": 720.0, "fps": 30, "mode": "Spectrum - Linear",
           "color": [0.1, 0.2, 0.3], "sensitivity": "1.5"}
    kw = worker.exporter_kwargs(cfg)
    assert kw == {"audio_path": "a.wav", "width": 1280, "height": 720, "fps": 30, "mode": "Spectrum - Linear",
                  "color": (0.1, 0.2, 0.3), "sensitivity": 1.5, "prefer_hevc": False, "view_state": {}, "gpu_device": ""}
    kw = worker.exporter_kwargs(dict(cfg, prefer_hevc=1, view_state=None, gpu_device=None))
    assert kw["prefer_hevc"] is True and kw["view_state"] == {} and kw["gpu_device"] == ""
    calls = []

    class FakeExporter:
        def __init__(self, **kw):
            calls.append(kw)

        def render_to_file(self, out_path, progress_cb=None):
            progress_cb(0)
            progress_cb(49.9)
            if out_path == "fail.mp4":
                raise RuntimeError("ffmpeg failed (exit 1):\nnope")
            progress_cb(100)

    monkeypatch.setattr(exporter, "Exporter", FakeExporter)
    p = tmp_path / "cfg.json"
    p.write_text(json.dumps(cfg))
    assert worker.main(["worker.py", str(p)]) == 0
    out = [json.loads(l) for l in capsys.readouterr().out.strip().splitlines()]
    assert out == [{"type": "progress", "pct": 0}, {"type": "progress", "pct": 49}, {"type": "progress", "pct": 100},
                   {"type": "done"}]
    p.write_text(json.dumps(dict(cfg, out_path="fail.mp4")))
    assert worker.main(["worker.py", str(p)]) == 1
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1]) == {"type": "error",
                                                                          "message": "ffmpeg failed (exit 1):\nnope"}
    del cfg["fps"]
    p.write_text(json.dumps(cfg))
    assert worker.main(["worker.py", str(p)]) == 1                      # KeyError inside the export block
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])["type"] == "error"


def test_synth_shapes():
    a = synth.sweep(4410, 44100)
    b = synth.pink(4800, 2)
    assert a.shape == (4410, 2) and a.dtype == np.float32 and abs(float(np.max(np.abs(a))) - 0.8) < 1e-3
    assert b.shape == (4800, 2) and abs(float(np.max(np.abs(b[:, 0]))) - 0.8) < 1e-6


# ------------------------------------------------------------------ radial raster oracle == brute force

def _brute_closed_stroke(V, w, h):
    """Pixel centre within the closed stroke outline, straight from the rule: rectangles + outer bevel triangles."""
    img = np.zeros((h, w), dtype=bool)
    py, px = np.mgrid[0:h, 0:w]
    cx, cy = px + 0.5, py + 0.5
    n = len(V)
    U = []
    for s in range(n):
        (x0, y0), (x1, y1) = V[s], V[(s + 1) % n]
        L = np.hypot(x1 - x0, y1 - y0)
        ux, uy = (x1 - x0) / L, (y1 - y0) / L
        U.append((ux, uy))
        along = (cx - x0) * ux + (cy - y0) * uy
        across = -(cx - x0) * uy + (cy - y0) * ux
        img |= (along >= 0) & (along <= L) & (np.abs(across) <= 1.0)
    for v in range(n):
        (uxa, uya), (uxb, uyb) = U[(v - 1) % n], U[v]
        cr = uxa * uyb - uya * uxb
        if cr == 0:
            continue
        sg = -1.0 if cr > 0 else 1.0
        p0 = np.array(V[v], dtype=float)
        p1 = p0 + sg * np.array([-uya, uxa])
        p2 = p0 + sg * np.array([-uyb, uxb])

        def side(a, b):
            return (b[0] - a[0]) * (cy - a[1]) - (b[1] - a[1]) * (cx - a[0])

        d0, d1, d2 = side(p0, p1), side(p1, p2), side(p2, p0)
        img |= ((d0 >= 0) & (d1 >= 0) & (d2 >= 0)) | ((d0 <= 0) & (d1 <= 0) & (d2 <= 0))
    return img


@pytest.mark.parametrize("seed", [0, 1])
def test_closed_stroke_and_polygon_masks_equal_brute_force(seed):
    rng = np.random.default_rng(seed)
    w, h, n = 220, 160, 90
    th = np.linspace(0, 2 * np.pi, n, endpoint=False)
    r = 45 + 18 * rng.random(n) * (1 + seed)
    V = np.stack([w / 2 + 0.137 + np.cos(th) * r, h / 2 + 0.291 + np.sin(th) * r], axis=1)
    got = R.unpack_mask(R.stroke_mask_closed(V, w, h), w)
    assert np.array_equal(got, _brute_closed_stroke(V, w, h))
    # even-odd polygon fill vs matplotlib-free winding test (crossing number along +x rays at pixel centres)
    poly = R.unpack_mask(R.polygon_mask(V, w, h), w)
    py, px = np.mgrid[0:h, 0:w]
    cx, cy = px + 0.5, py + 0.5
    inside = np.zeros((h, w), dtype=bool)
    for e in range(n):
        (x0, y0), (x1, y1) = V[e], V[(e + 1) % n]
        cond = (y0 <= cy) != (y1 <= cy)
        with np.errstate(divide="ignore", invalid="ignore"):
            xint = x0 + (cy - y0) * (x1 - x0) / (y1 - y0)
        inside ^= cond & (xint <= cx)
    assert np.array_equal(poly, inside)


# ------------------------------------------------------------------ round-1 additions: circular waveform, wire, blend

def test_wave_circular_angle_table_matches_oracle_expression():
    from oracle import radial_ref as RR
    for rot in (0.0, 33.0, -120.5):
        cs = engine.wave_circular_angle_table(rot)
        assert cs.shape == (1024, 2) and cs.dtype == np.float64
        samples = np.zeros(1024, np.float32)
        # r is constant for zero samples, so the oracle's path is (cx + cos a_i r, cy + sin a_i r)
        path = RR.wave_circular_path(samples, 0.25, 1000, 562, rot)
        r = min(1000, 562) * 0.28
        assert np.array_equal(path[:, 0], 1000 / 2.0 + cs[:, 0] * r)
        assert np.array_equal(path[:, 1], 562 / 2.0 + cs[:, 1] * r)


def test_yuv420p_oracle_known_answers():
    from oracle import wire_ref as WR
    def solid(rgb):
        f = np.empty((4, 8, 3), np.uint8)
        f[:] = rgb
        out = WR.rgb24_to_yuv420p(f)
        assert out.shape == (4 * 8 * 3 // 2,)
        return int(out[0]), int(out[32]), int(out[40])
    assert solid((0, 0, 0)) == (16, 128, 128)            # BT.601 limited range black
    assert solid((255, 255, 255)) == (235, 128, 128)     # white
    assert solid((255, 0, 0)) == (82, 90, 240)           # the classic 8-bit integer red
    assert solid((0, 0, 255)) == (41, 240, 110)
    y, cb, cr = solid((13, 13, 18))                      # Aurora's background colour (rt_widget.py:829)
    assert (y, cb, cr) == (28, 130, 128)              # (66*13 + 129*13 + 25*18 + 128 >> 8) + 16 = 12 + 16


def test_nv12_oracle_is_the_interleaved_yuv420p():
    from oracle import wire_ref as WR
    rng = np.random.default_rng(8)
    f = rng.integers(0, 256, size=(6, 8, 3), dtype=np.uint8)
    p, s = WR.rgb24_to_yuv420p(f), WR.rgb24_to_nv12(f)
    assert s.shape == p.shape and np.array_equal(s[:48], p[:48])
    assert np.array_equal(s[48::2], p[48:60]) and np.array_equal(s[49::2], p[60:])      # Cb Cr Cb Cr ...
    g = np.empty((2, 4, 3), np.uint8)
    g[:] = (255, 0, 0)
    assert WR.rgb24_to_nv12(g).tolist() == [82] * 8 + [90, 240, 90, 240]


def test_blend_rule_is_qt_premultiplied_source_over():
    # alpha 255 overwrites, alpha 0 keeps, and the six feather alphas over the disc colour give the documented value
    f = np.zeros((2, 2, 3), np.uint8)
    f[:] = (200, 100, 50)
    m = np.ones((2, 2), bool)
    R.blend_solid(f, m, (30, 30, 40), 255)
    assert (f == (30, 30, 40)).all()
    R.blend_solid(f, m, (13, 13, 18), 0)
    assert (f == (30, 30, 40)).all()
    for k in range(1, 7):
        R.blend_solid(f, m, R.BG, max(10, 70 - 10 * k))
    assert (f == (20, 20, 27)).all()


def test_new_entry_points_reject_bad_arguments_without_gpu():
    lib = _lib.load()
    # every call fails on the null context before touching a device
    assert lib.aur_rgb24_to_yuv420p(None, None, 1, 64, 64, None, None) != 0
    assert lib.aur_rgb24_to_nv12(None, None, 1, 64, 64, None, None) != 0
    assert lib.aur_raster_spans_nv12(None, None, 2, None, 1, 64, 64, None, None) != 0
    assert lib.aur_blend_planes(None, None, 6, b"\0" * 24, 1, 64, 64, None, None) != 0
    assert lib.aur_add_planes(None, None, 7, None, None, None, 0, 1, 64, 64, None, None) != 0
    assert lib.aur_wave_circular_vertices(None, None, None, 1, None, 64, 64, None, None) != 0
    assert lib.aur_feather_ring_vertices(None, None, None, None, 1, None, None, 64, 64, None, None) != 0
    assert b"null" in lib.aur_last_error()


def test_wide_translated_and_open_strokes_equal_brute_force():
    """stroke_mask_closed with a half width, a translation and open ends == pixel centres inside the union of the
    segment rectangles (square caps at open ends) and the bevel triangles, written directly from that rule."""
    rng = np.random.default_rng(3)
    w, h = 96, 80
    ang = np.sort(rng.uniform(0, 2 * np.pi, 9))
    rad = rng.uniform(12, 30, 9)
    V = np.stack([48 + rad * np.cos(ang), 40 + rad * np.sin(ang)], axis=1)
    py, px = np.mgrid[0:h, 0:w]
    cx, cy = px + 0.5, py + 0.5
    for closed in (True, False):
        for hw, dx, dy in ((1.0, 0.0, 0.0), (3.5, 4.25, -2.5)):
            got = R.unpack_mask(R.stroke_mask_closed(V, w, h, hw, dx, dy, closed), w)
            P = V + np.array([dx, dy])
            n = len(P)
            want = np.zeros((h, w), bool)
            nseg = n if closed else n - 1
            U = []
            for s in range(n):
                d = V[(s + 1) % n] - V[s]
                U.append(d / np.hypot(*d))
            for s in range(nseg):
                a, b, u = P[s], P[(s + 1) % n], U[s]
                L = np.hypot(*(b - a))
                along = (cx - a[0]) * u[0] + (cy - a[1]) * u[1]
                across = -(cx - a[0]) * u[1] + (cy - a[1]) * u[0]
                lo = -hw if (not closed and s == 0) else 0.0
                hi = L + (hw if (not closed and s == n - 2) else 0.0)
                want |= (along >= lo) & (along <= hi) & (np.abs(across) <= hw)
            for v in range(n):
                if not closed and v in (0, n - 1):
                    continue
                ua, ub = U[(v - 1) % n], U[v]
                cr = ua[0] * ub[1] - ua[1] * ub[0]
                sg = -1.0 if cr > 0 else 1.0
                p0 = P[v]
                p1 = p0 + sg * hw * np.array([-ua[1], ua[0]])
                p2 = p0 + sg * hw * np.array([-ub[1], ub[0]])

                def side(a, b):
                    return (b[0] - a[0]) * (cy - a[1]) - (b[1] - a[1]) * (cx - a[0])

                d0, d1, d2 = side(p0, p1), side(p1, p2), side(p2, p0)
                want |= ((d0 >= 0) & (d1 >= 0) & (d2 >= 0)) | ((d0 <= 0) & (d1 <= 0) & (d2 <= 0))
            diff = int(np.sum(got != want))
            assert diff <= 2, (closed, hw, diff)      # pixel centres within rounding of an outline edge
            assert got.sum() > 200


def test_round_stroke_equals_distance_to_polyline():
    """round_join=True (RoundCap / RoundJoin pen, the shadow of "Waveform - Linear", rt_widget.py:918-927) == every pixel
    centre within the half width of the translated polyline; pixels closer than 1e-9 to the boundary are not compared."""
    rng = np.random.default_rng(11)
    w, h = 96, 64
    xs = np.arange(0, w, 4, dtype=np.float64)
    V = np.stack([xs, 32 + rng.uniform(-20, 20, len(xs))], axis=1)
    py, px = np.mgrid[0:h, 0:w]
    cx, cy = px + 0.5, py + 0.5
    for hw, d in ((1.0, 0.0), (2.5, 3.0), (5.0, 8.0)):
        got = R.unpack_mask(R.stroke_mask_closed(V, w, h, hw, d, d, False, round_join=True), w)
        P = V + d
        dist = np.full((h, w), np.inf)
        for s in range(len(P) - 1):
            a, b = P[s], P[s + 1]
            ab = b - a
            t = np.clip(((cx - a[0]) * ab[0] + (cy - a[1]) * ab[1]) / (ab @ ab), 0.0, 1.0)
            dist = np.minimum(dist, np.hypot(cx - (a[0] + t * ab[0]), cy - (a[1] + t * ab[1])))
        sure = np.abs(dist - hw) > 1e-9
        assert np.array_equal(got[sure], (dist <= hw)[sure])
        assert got.any() and not got.all()


# ------------------------------------------------------------------ FFmpeg argv / encoder list == the reference's

def test_ffmpeg_argv_and_device_list_match_reference_matrix():
    """build_ffmpeg_cmd / list_gpu_export_devices (export/exporter.py:368-491) over 3 systems x 9 encoder sets x 16
    device strings x h264/hevc: argv, error messages and the dropdown list equal what the reference returned
    (tests/golden/ffmpeg_cmds.npz, made by make_golden.py's replay_ffmpeg_cases run on the reference module)."""
    import importlib.util
    import json
    import os
    from aurora_amd.export import exporter as mine
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.loads(str(np.load(os.path.join(here, "ffmpeg_cmds.npz"))["cases_json"]))
    got = json.loads(mg.replay_ffmpeg_cases(mine))
    assert got.keys() == want.keys() and len(want) == 864 + 27
    bad = [k for k in want if got[k] != want[k]]
    assert not bad, (bad[:5], got[bad[0]], want[bad[0]])


def test_ffmpeg_argv_with_the_opt_in_wire_formats(monkeypatch):
    """Only the input "-pix_fmt" changes with view_state["hip_wire_format"]; on the VAAPI path an nv12 input is what the
    reference's own filter chain "format=nv12,hwupload" (export/exporter.py:464) converts to."""
    from aurora_amd.export import exporter as mine
    monkeypatch.setattr(mine, "_ffmpeg_bin", lambda: "ffmpeg")
    monkeypatch.setattr(mine, "_ffmpeg_encoders", lambda ffmpeg: {"libx264", "h264_vaapi", "hevc_vaapi"})
    monkeypatch.setattr(mine.platform, "system", lambda: "Linux")
    ref, fmt = mine.build_ffmpeg_cmd(1920, 1080, 60, "out", "a.wav", gpu_device="vaapi:/dev/dri/renderD128")
    assert fmt == "rgb24" and ref[ref.index("-vf") + 1] == "format=nv12,hwupload" and ref[-1] == "out.mp4"
    for wire in ("yuv420p", "nv12"):
        cmd, fmt = mine.build_ffmpeg_cmd(1920, 1080, 60, "out", "a.wav", gpu_device="vaapi:/dev/dri/renderD128",
                                         input_pix_fmt=wire)
        assert fmt == wire
        i = cmd.index("-pix_fmt")
        assert cmd[i + 1] == wire and cmd[:i + 1] == ref[:i + 1] and cmd[i + 2:] == ref[i + 2:]
    with pytest.raises(ValueError):
        mine.build_ffmpeg_cmd(64, 64, 30, "out", "a.wav", input_pix_fmt="nv21")


class _DuckRenderer:
    """The duck type `_run_pipe` accepts (exporter.py:536-540): .duration and .render_time(t)."""
    def __init__(self, duration, shape):
        self.duration, self.shape, self.times = duration, shape, []

    def render_time(self, t):
        self.times.append(t)
        return np.full(self.shape, len(self.times), np.uint8)[:, ::1]


def test_run_pipe_matches_reference_loop(tmp_path):
    """Exporter._run_pipe (exporter.py:531-567) with a stand-in child process: frame bytes in order, progress every
    max(1, int(fps / 2)) frames as int(i * 100 / total) and a final 100, non-zero exit -> RuntimeError with stderr,
    a child that stops reading ends the loop quietly."""
    import sys
    from aurora_amd.export.exporter import Exporter
    ex = Exporter("unused.wav", width=4, height=2, fps=4, mode="Spectrum - Linear")
    out = tmp_path / "frames.bin"
    sink_cmd = [sys.executable, "-c", f"import sys; open({str(out)!r}, 'wb').write(sys.stdin.buffer.read())"]
    r = _DuckRenderer(2.1, (2, 4, 3))
    seen = []
    ex._run_pipe(r, sink_cmd, progress_cb=seen.append)
    assert r.times == [i / 4 for i in range(9)]                        # ceil(2.1 * 4) frames at t = i / fps
    assert out.read_bytes() == b"".join(bytes([k]) * 24 for k in range(1, 10))
    assert seen == [0, 22, 44, 66, 88, 100]
    st = ex.last_stats                                                 # the export's own figures (hip_stats_path)
    assert st["frames"] == 9 and st["bytes_per_frame"] == 24 and st["frame_range"] == [0, 9] and st["wire_format"] == "rgb24"
    assert st["pipe"] == {"spliced_bytes": 0, "copied_bytes": 9 * 24, "path": "write"} and st["seconds"] > 0
    # a progress callback that raises is ignored (:546-549)
    ex._run_pipe(_DuckRenderer(0.5, (2, 4, 3)), sink_cmd, progress_cb=lambda p: 1 / 0)
    fail_cmd = [sys.executable, "-c", "import sys; sys.stdin.buffer.read(); sys.stderr.write('boom'); sys.exit(3)"]
    with pytest.raises(RuntimeError) as ei:
        ex._run_pipe(_DuckRenderer(0.5, (2, 4, 3)), fail_cmd)
    assert str(ei.value) == "ffmpeg failed (exit 3):\nboom"
    # the child exits without reading: BrokenPipeError stops the writes, exit code 0 -> no error
    big = Exporter("unused.wav", width=1024, height=512, fps=30, mode="Spectrum - Linear")
    quit_cmd = [sys.executable, "-c", "import sys; sys.exit(0)"]
    r = _DuckRenderer(3.0, (512, 1024, 3))
    seen = []
    big._run_pipe(r, quit_cmd, progress_cb=seen.append)
    assert len(r.times) < 90 and seen[-1] == 100


def test_run_pipe_survives_a_chatty_child(tmp_path):
    """A child that writes more to stderr / stdout than a pipe holds (FFmpeg's statistics over a long encode) while it
    still reads frames: both pipes are drained during the export, and the tail of stderr reaches the error message."""
    import sys
    from aurora_amd.export.exporter import Exporter
    ex = Exporter("unused.wav", width=64, height=32, fps=30, mode="Spectrum - Linear")
    chatty = [sys.executable, "-c",
              "import sys\n"
              "for i in range(60):\n"
              "    sys.stdin.buffer.read(64 * 32 * 3)\n"
              "    sys.stderr.write('x' * 8192 + '\\n'); sys.stderr.flush()\n"
              "    sys.stdout.write('y' * 8192 + '\\n'); sys.stdout.flush()\n"
              "sys.stdin.buffer.read(); sys.stderr.write('last words'); sys.exit(5)"]
    with pytest.raises(RuntimeError) as ei:
        ex._run_pipe(_DuckRenderer(2.0, (32, 64, 3)), chatty)
    msg = str(ei.value)
    assert msg.startswith("ffmpeg failed (exit 5):") and msg.endswith("last words") and len(msg) <= (1 << 16) + 64


def test_load_audio_wav_variants_and_ffmpeg_fallback(tmp_path, monkeypatch):
    """Audio ingest (exporter.py:108-113 with `_FFmpegDecodedAudio` :46-81): WAV files are read whole as float32
    [n, C]; anything else goes through `ffmpeg ... -f f32le -ac 1 -ar 48000 -` (a stand-in script here), with the
    reference's two error messages."""
    import os
    import stat
    import sys
    import wave
    from scipy.io import wavfile
    from aurora_amd.export import exporter as E
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, (1000, 2)).astype(np.int16)
    with wave.open(str(tmp_path / "a16.wav"), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
    a, sr = E.load_audio(str(tmp_path / "a16.wav"))
    assert sr == 44100 and a.dtype == np.float32 and a.shape == (1000, 2)
    assert np.array_equal(a, pcm.astype(np.float32) / 32768.0)
    f32 = (rng.standard_normal((777, 1)) * 0.1).astype(np.float32)
    wavfile.write(str(tmp_path / "f32.wav"), 48000, f32[:, 0])
    a, sr = E.load_audio(str(tmp_path / "f32.wav"))
    assert sr == 48000 and a.shape == (777, 1) and np.array_equal(a, f32)
    # 24-bit PCM: sample / 2^23 whichever reader takes it
    v = rng.integers(-(1 << 23), 1 << 23, 300).astype(np.int32)
    raw = b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v)
    with wave.open(str(tmp_path / "a24.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(3); w.setframerate(48000); w.writeframes(raw)
    a, sr = E.load_audio(str(tmp_path / "a24.wav"))
    assert a.shape == (300, 1) and np.array_equal(a[:, 0], v.astype(np.float32) / 8388608.0)
    assert np.array_equal(E._read_wav_stdlib(str(tmp_path / "a24.wav"))[0], a)
    # not a WAV: the FFmpeg decode, mono at 48 kHz
    mp3 = tmp_path / "song.mp3"
    mp3.write_bytes(b"ID3 not really audio")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    pcm_out = tmp_path / "decoded.f32"
    want = (rng.standard_normal(4321) * 0.2).astype(np.float32)
    pcm_out.write_bytes(want.tobytes())
    args_out = tmp_path / "args.txt"
    script = bindir / "ffmpeg"
    script.write_text(f"#!{sys.executable}\nimport sys\nopen({str(args_out)!r}, 'w').write('\\n'.join(sys.argv[1:]))\n"
                      f"sys.stdout.buffer.write(open({str(pcm_out)!r}, 'rb').read())\n")
    script.chmod(script.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ.get("PATH", ""))
    a, sr = E.load_audio(str(mp3))
    assert sr == 48000 and a.shape == (4321, 1) and np.array_equal(a[:, 0], want)
    assert args_out.read_text().split("\n") == ["-v", "error", "-i", str(mp3), "-f", "f32le", "-acodec", "pcm_f32le",
                                                 "-ac", "1", "-ar", "48000", "-"]                  # exporter.py:54-58
    script.write_text(f"#!{sys.executable}\nimport sys\nsys.stderr.write('bad input')\nsys.exit(1)\n")
    with pytest.raises(RuntimeError) as ei:
        E.load_audio(str(mp3))
    assert str(ei.value) == "ffmpeg decode failed: bad input"
    monkeypatch.setattr(E.shutil, "which", lambda name: None)
    with pytest.raises(RuntimeError) as ei:
        E.load_audio(str(mp3))
    assert str(ei.value) == "ffmpeg not found for decoding non-WAV formats."


def test_frame_compare_worker_in_spawned_pool():
    """The worker of tests/test_gpu_e2e_parity.py (oracle render + compare through shared memory, spawned processes)."""
    import multiprocessing as mp
    from multiprocessing import shared_memory
    from _frame_compare_worker import compare_frame
    from oracle import raster_ref as R
    W, H = 320, 180
    rng = np.random.default_rng(0)
    spec = rng.random(1025).astype(np.float32)
    smp = rng.random(1024).astype(np.float32) - 0.5
    fr = R.render_frame("Spectrum - Linear", smp, spec, 0.3, (1, 2, 3), W, H)
    fb = W * H * 3
    shm = shared_memory.SharedMemory(create=True, size=2 * fb)
    try:
        v = np.ndarray((2, H, W, 3), np.uint8, buffer=shm.buf)
        v[0] = fr
        v[1] = fr
        v[1, 100, 100] = (9, 9, 9)
        del v
        with mp.get_context("spawn").Pool(2) as p:
            res = p.map_async(compare_frame, [(shm.name, j * fb, "Spectrum - Linear", smp, spec, 0.3, (1, 2, 3), W, H)
                                              for j in range(2)]).get(timeout=60)
    finally:
        shm.close()
        shm.unlink()
    assert res[0] == (0, 0, float("inf"))
    assert res[1][0] == 1 and res[1][1] == 1 and res[1][2] > 40


def test_numpy_cabs_formula():
    """np.abs(complex64) on this host == max * sqrt(fma(d, d, 1)), d = min / max in float32 -- the formula
    csrc/stft.hip:cabs_c64 evaluates for the displayed bins.  numpy's AVX2 / AVX-512 loops use it when FMA3 is there;
    a host without FMA3 takes another path, and then the fixtures (generated on an FMA3 host) are the pin, not this test."""
    from numpy._core._multiarray_umath import __cpu_features__ as feat
    if not (feat.get("FMA3") and feat.get("AVX2")):
        pytest.skip("host numpy has no AVX2 + FMA3 dispatch")
    rng = np.random.default_rng(0)
    n = 2_000_000
    z = (rng.standard_normal(n) * 10 ** rng.uniform(-6, 3, n)
         + 1j * rng.standard_normal(n) * 10 ** rng.uniform(-6, 3, n)).astype(np.complex64)
    z[:4] = [0, 1, 1j, 3 + 4j]
    f32 = np.float32
    a, b = np.abs(z.real), np.abs(z.imag)
    mx, mn = np.maximum(a, b), np.minimum(a, b)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.where(mx > 0, mn / mx, 0).astype(f32)
    fma = (d.astype(np.float64) * d.astype(np.float64) + 1.0).astype(f32)     # exact product, one rounding
    want = (mx * np.sqrt(fma)).astype(f32)
    assert np.array_equal(np.abs(z), want)


def test_ffmpeg_pipe_is_grown():
    """`_run_pipe` asks for a 1 MiB pipe (a 4K frame is 24.9 MB; the default pipe holds 64 KB)."""
    import fcntl
    import subprocess
    import sys
    from aurora_amd.export.exporter import _grow_pipe
    proc = subprocess.Popen([sys.executable, "-c", "import sys; sys.stdin.buffer.read()"], stdin=subprocess.PIPE)
    try:
        got = _grow_pipe(proc.stdin)
        size = fcntl.fcntl(proc.stdin.fileno(), getattr(fcntl, "F_GETPIPE_SZ", 1032))
        assert got == 0 or size >= got >= 65536
    finally:
        proc.stdin.close()
        proc.wait()
    assert _grow_pipe(object()) == 0                 # anything that is not a pipe: left alone


def test_image_decode_falls_back_to_ffmpeg(tmp_path, monkeypatch):
    """load_image_rgba without Pillow: one raw rgba frame from FFmpeg (a stand-in script here), its size taken from the
    stream description on stderr and checked against the bytes that arrived; anything else is an error."""
    import stat
    import sys
    from aurora_amd import images
    rgba = (np.arange(7 * 5 * 4) % 251).astype(np.uint8).reshape(5, 7, 4)
    raw = tmp_path / "frame.raw"
    raw.write_bytes(rgba.tobytes())
    ff = tmp_path / "ffmpeg"
    ff.write_text(f"""#!{sys.executable}
import sys
a = sys.argv[1:]
if "-encoders" in a:
    sys.exit(0)
src = a[a.index("-i") + 1]
if src.endswith("bad.png"):
    sys.stderr.write("bad.png: Invalid data found when processing input\\n"); sys.exit(1)
sys.stderr.write("Input #0, png_pipe, from 'x.png':\\n  Duration: N/A, bitrate: N/A\\n"
                 "  Stream #0:0: Video: png, rgba(pc, gbr/unknown/unknown), 7x5 [SAR 2835:2835 DAR 7:5], 25 fps\\n")
assert a[a.index("-pix_fmt") + 1] == "rgba" and a[-1] == "-"
sys.stdout.buffer.write(open({str(raw)!r}, "rb").read() if not src.endswith("short.png") else b"1234")
""")
    ff.chmod(ff.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("AURORA_FFMPEG", str(ff))
    monkeypatch.setitem(sys.modules, "PIL", None)                      # `from PIL import Image` raises ImportError
    for name in ("x.png", "bad.png", "short.png"):
        (tmp_path / name).write_bytes(b"not really a png")
    got = images.load_image_rgba(str(tmp_path / "x.png"))
    assert got.dtype == np.uint8 and np.array_equal(got, rgba)
    with pytest.raises(RuntimeError, match="could not decode"):
        images.load_image_rgba(str(tmp_path / "bad.png"))
    with pytest.raises(RuntimeError, match="did not report a size"):
        images.load_image_rgba(str(tmp_path / "short.png"))
    with pytest.raises(FileNotFoundError):
        images.load_image_rgba(str(tmp_path / "missing.png"))


def test_colour_strings_as_qcolor_reads_them():
    """view_state colours (`radial_fill_color`, `glow_color`): the '#' forms QColor accepts; anything else is an invalid
    QColor, which the reference's setters ignore (rt_widget.py:369-372), i.e. the default stays."""
    from aurora_amd.engine import FillConfig
    fb = (1, 2, 3, 4)
    assert FillConfig.parse_colour("#80FFFFFF", fb) == (255, 255, 255, 0x80)
    assert FillConfig.parse_colour("#c0ff8040", fb) == (0xff, 0x80, 0x40, 0xc0)
    assert FillConfig.parse_colour("#FF8040", fb) == (0xff, 0x80, 0x40, 255)
    assert FillConfig.parse_colour("#F84", fb) == (0xff, 0x88, 0x44, 255)
    assert FillConfig.parse_colour("#FFF888444", fb) == (0xff, 0x88, 0x44, 255)
    assert FillConfig.parse_colour("#FFFF88884444", fb) == (0xff, 0x88, 0x44, 255)
    for bad in ("", "#", "#12", "#GGHHII", "FF8040", "#12345", None, 17):
        assert FillConfig.parse_colour(bad, fb) == fb
    assert FillConfig.from_view_state({"radial_fill_color": "nonsense"}).rgba == (255, 255, 255, 48)   # rt_widget.py:102
    assert FillConfig.from_view_state({"radial_fill_threshold": "x"}).threshold == 0.1


@pytest.mark.parametrize("width", [1, 2, 3, 4])
def test_stdlib_wav_reader_agrees_with_the_scipy_path(tmp_path, width):
    """The last-resort PCM reader (no soundfile, scipy refuses the file) scales 8 / 16 / 24 / 32-bit samples exactly as
    the scipy path of load_audio does (libsndfile's float32 conversion: x / 2^(bits-1), 8-bit unsigned around 128)."""
    import wave
    from aurora_amd.export import exporter
    rng = np.random.default_rng(width)
    n, ch = 257, 2
    if width == 1:
        raw = rng.integers(0, 256, size=n * ch, dtype=np.uint8).tobytes()
    else:
        v = rng.integers(-(1 << (8 * width - 1)), 1 << (8 * width - 1), size=n * ch, dtype=np.int64)
        v[:3] = [-(1 << (8 * width - 1)), (1 << (8 * width - 1)) - 1, 0]
        raw = b"".join(int(x).to_bytes(width, "little", signed=True) for x in v)
    path = str(tmp_path / f"pcm{8 * width}.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(ch)
        w.setsampwidth(width)
        w.setframerate(22050)
        w.writeframes(raw)
    a, sr = exporter._read_wav_stdlib(path)
    b, sr2 = exporter.load_audio(path)
    assert sr == sr2 == 22050 and a.shape == b.shape == (n, ch) and a.dtype == b.dtype == np.float32
    assert np.array_equal(a, b)
    assert -1.0 <= a.min() and a.max() <= 1.0          # 32-bit full scale rounds to 1.0 in float32


def test_every_entry_point_rejects_a_null_context():
    """No GPU needed: every entry point that takes a context returns an error code (and sets a message) for a null context
    and null / zero arguments instead of touching them."""
    import ctypes as C
    lib = _lib.load()
    free_standing = {"aur_version", "aur_last_error", "aur_device_count", "aur_create", "aur_destroy", "aur_host_alloc",
                     "aur_host_free", "aur_raster_work_bytes", "aur_glow_work_bytes", "aur_device_pci_bus_id"}
    checked = 0
    for name in _lib.SYMBOLS:
        if name in free_standing:
            continue
        fn = getattr(lib, name)
        args = []
        for t in fn.argtypes:
            if t in (C.c_double, C.c_float):
                args.append(0.0)
            elif t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
                args.append(None)
            else:
                args.append(0)
        assert fn(*args) != 0, name
        assert lib.aur_last_error(), name
        checked += 1
    assert checked >= 45
    out = C.c_size_t(0)
    assert lib.aur_raster_work_bytes(4, 3840, C.byref(out)) == 0 and out.value > 0
    assert lib.aur_raster_work_bytes(4, 3840, None) != 0
    assert lib.aur_host_alloc(None, 16) != 0 and lib.aur_host_free(None) == 0


# ---- the pipe into FFmpeg: vmsplice of whole ring slots (exporter._PipeSink) -----------------------------------------------

def _drain_child(extra=()):
    import subprocess
    return subprocess.Popen([sys.executable, os.path.join(REPO, "tools", "pipe_drain.py"), "--sha256", *extra],
                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=10 ** 7)


_CH = 32 << 20        # the sink's staging chunk (exporter._PipeSink.CHUNK): sizes around one, two and three chunks
_PIPE_SIZES = (24_000_011, 5, 4 * (1 << 20), 4 * (1 << 20) - 1, 1_000_003, 0, _CH, _CH + 1, _CH + (1 << 20) - 1,
               _CH + (1 << 20), 2 * _CH + 100, 17_123_457, 2 * _CH + (1 << 20) + 7, 70_000_003)


@pytest.mark.parametrize("path", ["plain", "direct", "staged"])
def test_pipe_sink_stream_is_the_written_bytes_even_when_buffers_are_reused(path):
    """A slow reader (4 KiB reads + sha256) keeps the pipe full; the writer overwrites its buffer the moment `write`
    returns, as the pinned ring does.  The child must still see exactly the bytes that were handed over: on the direct
    path spliced pages have left the pipe by the time `write` returns (the tail rule), on the staged path the caller's
    memory is never in the pipe and a staging buffer is reused only after a whole later chunk went in behind it."""
    import hashlib
    from aurora_amd.export.exporter import _PipeSink, _grow_pipe
    assert _PipeSink.CHUNK == _CH
    child = _drain_child(("--chunk", "4096"))
    cap = _grow_pipe(child.stdin)
    sink = _PipeSink(child.stdin, splice=path != "plain", staged=path == "staged")
    rng = np.random.default_rng(5)
    buf = np.empty(max(_PIPE_SIZES), np.uint8)
    want, total = hashlib.sha256(), 0
    for n in _PIPE_SIZES:
        part = buf[:n]
        part[:] = rng.integers(0, 256, size=n, dtype=np.uint8)
        want.update(part.tobytes())
        total += n
        sink.write(memoryview(part))
        buf[:] = 0xEE                                   # the slot is reused at once
    child.stdin.close()
    rep = json.loads(child.stdout.read())
    sink.close()
    assert child.wait() == 0
    assert rep["bytes"] == total and rep["sha256"] == want.hexdigest()
    big = [n for n in _PIPE_SIZES if cap and n >= 4 * cap]
    if path == "plain" or sink._vmsplice is None or not cap:
        assert sink.spliced_bytes == 0 and sink.copied_bytes == total
    elif path == "direct":
        assert sink.spliced_bytes == sum(n - cap for n in big)      # all but the last `capacity` bytes of a big buffer
        assert sink.spliced_bytes + sink.copied_bytes == total
    else:
        assert sink.spliced_bytes == sum(big) and sink.copied_bytes == total - sum(big)


def test_pipe_sink_broken_pipe_and_refused_memory(monkeypatch):
    import subprocess
    from aurora_amd.export import exporter as E
    # the reader goes away: BrokenPipeError, which the pipeline treats as "stop writing" (exporter.py:543-544)
    for staged in (False, True):
        child = subprocess.Popen([sys.executable, "-c", "import sys; sys.stdin.buffer.read(1 << 16)"], stdin=subprocess.PIPE)
        E._grow_pipe(child.stdin)
        sink = E._PipeSink(child.stdin, splice=True, staged=staged)
        big = np.zeros(64 << 20, np.uint8)
        with pytest.raises(BrokenPipeError):
            for _ in range(8):
                sink.write(memoryview(big))
        sink.close()
        child.wait()
    # memory the kernel will not splice (EFAULT, as for hipHostMalloc memory): the sink stages it from then on
    child = _drain_child()
    E._grow_pipe(child.stdin)
    sink = E._PipeSink(child.stdin, splice=True)
    data = np.arange(40 << 20, dtype=np.uint8)
    lo, hi = data.ctypes.data, data.ctypes.data + data.nbytes
    real, refused = sink._vmsplice, []

    def picky(fd, iov, n, flags):
        base = ctypes.cast(iov, ctypes.POINTER(E._IoVec)).contents.iov_base
        if lo <= base < hi:
            refused.append(base)
            ctypes.set_errno(14)                              # EFAULT
            return -1
        return real(fd, iov, n, flags)

    monkeypatch.setattr(sink, "_vmsplice", picky)
    import hashlib
    sink.write(memoryview(data))
    sink.write(memoryview(data))
    child.stdin.close()
    rep = json.loads(child.stdout.read())
    sink.close()
    child.wait()
    assert rep["bytes"] == 2 * data.nbytes and rep["sha256"] == hashlib.sha256(data.tobytes() * 2).hexdigest()
    assert len(refused) == 1 and sink.refused_errno == 14 and sink.spliced_bytes == 2 * data.nbytes
    # AUR_PIPE_SPLICE=0 turns it off
    monkeypatch.setenv("AUR_PIPE_SPLICE", "0")
    child = _drain_child()
    assert E._PipeSink(child.stdin)._fd is None
    child.stdin.close()
    child.wait()
