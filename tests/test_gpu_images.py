"""GPU parity of the antialiased centre disc, the centre image and the background image (csrc/image.hip) against the
oracle's statement of the same rules (oracle/raster_ref.py: coverage_aa, smooth_scale_pixel, center_image_placement,
compose_background).  Qt's rasteriser / scaler are not in the reference tree: the rules are "parity unpinned" against
Qt; the reference's own draw calls for these features are pinned in tests/golden/images_*.npz (test_oracle_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import raster_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from aurora_amd.engine import get_engine
    return get_engine(0)


def _np(t):
    return t.cpu().numpy()


def _coverage(eng, verts, W, H, box=None):
    """Raw coverage of a polygon: an opaque white image under the clip leaves byte_mul(255, cov) = cov in the frame."""
    import torch
    from aurora_amd import _lib
    v = torch.from_numpy(np.ascontiguousarray(verts, np.float64)[None]).to(eng.device)
    white = torch.full((H, W), -1, dtype=torch.int32, device=eng.device)
    pl = torch.tensor([[W, H, 0, 0]], dtype=torch.int32, device=eng.device)
    frames = torch.zeros((1, H, W, 3), dtype=torch.uint8, device=eng.device)
    full = torch.zeros((1, H, (W + 31) // 32), dtype=torch.int32, device=eng.device)
    src = _lib.DiscSource(2, 0, white.data_ptr(), W, H, pl.data_ptr(), 1)
    eng.disc_composite(v, True, 1, W, H, box or (0, 0, W, H), frames=frames, source=src, full_plane=full)
    f = _np(frames)[0]
    assert (f[..., 0] == f[..., 1]).all() and (f[..., 0] == f[..., 2]).all()
    return f[..., 0], R.unpack_mask(_np(full)[0].view(np.uint32), W)


def _envelope_like(W, H, seed, wavy=0.15):
    rng = np.random.default_rng(seed)
    n = 361
    a = 2 * np.pi * np.arange(n) / 360.0 - np.pi / 2
    r = 0.3 * min(W, H) * (1.0 + wavy * np.sin(3 * a + rng.uniform(0, 6)) + 0.05 * rng.standard_normal(n))
    return np.stack([W / 2 + 0.37 + np.cos(a) * r, H / 2 - 0.21 + np.sin(a) * r], axis=1)


@pytest.mark.parametrize("size", [(256, 160), (640, 360), (1000, 562), (3840, 96)])
def test_aa_coverage_bit_exact(eng, size):
    W, H = size
    polys = [_envelope_like(W, H, 1), _envelope_like(W, H, 2, wavy=0.6),
             np.array([[10.5, 10.25], [W - 20.5, 10.25], [W - 20.5, H - 15.0], [10.5, H - 15.0]]),
             np.array([[W / 2, 3.1], [W / 2 + 40.3, H - 4.2], [W / 2 - 37.9, H - 9.6]]),
             np.array([[-30.0, -20.0], [W + 25.0, H / 2.0], [-10.0, H + 40.0]]),          # reaches outside the frame
             np.array([[-2500.0, H / 3.0], [W / 2.0, 5.5], [W + 1800.0, H - 7.25], [W / 3.0, H - 3.0]])]   # far outside
    for k, V in enumerate(polys):
        got, full = _coverage(eng, V, W, H)
        want = R.coverage_aa(V, W, H)
        assert np.array_equal(got, want), (size, k, int(np.sum(got != want)))
        assert np.array_equal(full, want == 255), (size, k)
    # a box narrower than the frame: same coverage inside it, nothing outside
    V = polys[0]
    x0, y0 = int(V[:, 0].min()) - 2, int(V[:, 1].min()) - 2
    x1, y1 = int(V[:, 0].max()) + 3, int(V[:, 1].max()) + 3
    x0, y0, x1, y1 = max(0, x0), max(0, y0), min(W, x1), min(H, y1)
    got, _ = _coverage(eng, V, W, H, box=(x0, y0, x1 - x0, y1 - y0))
    assert np.array_equal(got, R.coverage_aa(V, W, H))


@pytest.mark.parametrize("shape", [((8, 6), (8, 6)), ((8, 6), (21, 17)), ((40, 30), (13, 7)), ((9, 33), (27, 11)),
                                   ((31, 5), (6, 19)), ((64, 48), (63, 49)), ((3, 3), (1, 1)), ((1, 1), (5, 4))])
def test_smooth_scale_bit_exact(eng, shape):
    import torch
    from aurora_amd.images import premultiply_argb
    (sw, sh), (dw, dh) = shape
    rng = np.random.default_rng(sw * 131 + dh)
    rgba = rng.integers(0, 256, size=(sh, sw, 4), dtype=np.uint8)
    if (sw + dw) % 2:
        rgba[..., 3] = 255
    pm = R.premultiply(rgba)
    packed = premultiply_argb(rgba)
    chk = ((pm[..., 3].astype(np.uint32) << 24) | (pm[..., 0].astype(np.uint32) << 16) |
           (pm[..., 1].astype(np.uint32) << 8) | pm[..., 2].astype(np.uint32))
    assert np.array_equal(packed, chk)
    img = eng.upload_image(packed)
    out = _np(eng.scale_image(img, dw, dh)).view(np.uint32)
    want = R.smooth_scale(pm, dw, dh)
    got = np.stack([(out >> 16) & 255, (out >> 8) & 255, out & 255, out >> 24], axis=-1).astype(np.uint8)
    assert np.array_equal(got, want), (shape, int(np.sum(got != want)))


def test_center_image_placement_matches_oracle(eng):
    import torch
    from aurora_amd import _lib
    rng = np.random.default_rng(3)
    F, W, H = 64, 1280, 720
    env = np.stack([_envelope_like(W, H, 10 + f) for f in range(F)])
    energy = rng.uniform(0, 1, F)
    phase = rng.uniform(0, 2 * np.pi, F)
    d_env = torch.from_numpy(env).to(eng.device)
    d_e, d_p = torch.from_numpy(energy).to(eng.device), torch.from_numpy(phase).to(eng.device)
    for (iw, ih, zoom, cm) in ((640, 480, 100, 0), (333, 517, 180, 35), (4000, 3000, 50, 100), (17, 9, 250, 60)):
        out = torch.empty((F, 4), dtype=torch.int32, device=eng.device)
        _lib.check(eng.lib.aur_center_image_placement(eng.ctx.handle, C.c_void_p(d_env.data_ptr()), 361, 0,
                                                      C.c_void_p(d_e.data_ptr()), C.c_void_p(d_p.data_ptr()), F, iw, ih,
                                                      zoom, cm, C.c_void_p(out.data_ptr()), eng.stream()), "placement")
        got = _np(out)
        for f in range(F):
            want = R.center_image_placement(env[f], iw, ih, zoom, cm, float(energy[f]), float(phase[f]))
            assert tuple(int(v) for v in got[f]) == want, (iw, ih, zoom, cm, f)


def _write_png(path, rgba):
    from PIL import Image
    Image.fromarray(rgba, "RGBA").save(path)


def _test_image(w, h, seed, alpha=False):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(1, w - 1)), (yy * 255 // max(1, h - 1)), ((xx + yy) * 7) % 256,
                    np.full_like(xx, 255)], axis=-1).astype(np.uint8)
    img[..., :3] ^= rng.integers(0, 32, size=(h, w, 3), dtype=np.uint8)
    if alpha:
        img[..., 3] = (255 * (((xx // 5) + (yy // 5)) % 3) // 2).astype(np.uint8)
    return img


@pytest.mark.parametrize("case", ["static", "zoom", "motion", "wavy", "alpha"])
def test_radial_center_image_frames(eng, tmp_path, case):
    """"Spectrum - Radial" with a centre image: GPU frames == the oracle's frame (stroke, antialiased clip, smooth-scaled
    image at the reference's placement) for the frame-independent disc, the zoom, the motion jitter and a wavy disc."""
    pytest.importorskip("PIL")
    from aurora_amd import synth
    from aurora_amd.engine import RadialConfig
    from aurora_amd.pipeline import FramePipeline
    W, H, sr, fps = 320, 200, 48000, 60
    rgba = _test_image(97, 61, 5, alpha=(case == "alpha"))
    path = str(tmp_path / "centre.png")
    _write_png(path, rgba)
    vs = dict(center_image_path=path)
    if case == "zoom":
        vs["center_image_zoom"] = 170
    if case == "motion":
        vs.update(center_motion=80, center_image_zoom=60)
    if case == "wavy":
        vs.update(edge_waviness=70, feather_audio_enabled=True, feather_audio_amount=90, center_motion=30)
    cfg = RadialConfig.from_view_state(vs, 1.0)
    audio = synth.pink(3000, 2, seed=3)
    track = eng.upload_track(audio, sr, fps)
    pipe = FramePipeline(eng, track, "Spectrum - Radial", W, H, radial=cfg)
    an, sc = pipe.analyse()
    frames = _np(pipe.render_device())
    verts, env = eng.radial_vertices(pipe._radial_prep, an, sc, W, H, 0, an.n_frames)
    verts, env, pen = _np(verts), _np(env), _np(sc.pen_rgb)[:, :3]
    energy, phase = _np(sc.energy), _np(sc.phase)
    pm = R.premultiply(rgba)
    for f in range(an.n_frames):
        pl = R.center_image_placement(env[f], 97, 61, cfg.center_image_zoom, cfg.center_motion, float(energy[f]),
                                      float(phase[f]))
        want = R.render_radial_frame(verts[f], env[f], pen[f], W, H, image_pm=pm, placement=pl)
        assert np.array_equal(frames[f], want), (case, f, int(np.sum(frames[f] != want)))
    assert len({frames[f].tobytes() for f in range(an.n_frames)}) > 1


@pytest.mark.parametrize("seed", range(8))
def test_random_center_images(eng, tmp_path, seed):
    """Random frame size, image size (tiny to larger than the frame), zoom, motion, disc waviness and rotation, with and
    without an alpha channel: every frame against the oracle (antialiased clip x smooth-scaled image at the placement)."""
    pytest.importorskip("PIL")
    from aurora_amd import synth
    from aurora_amd.engine import RadialConfig
    from aurora_amd.pipeline import FramePipeline
    rng = np.random.default_rng(300 + seed)
    W, H = int(rng.integers(5, 12)) * 32 - int(rng.choice([0, 0, 7, 16])), int(rng.integers(100, 240))
    iw, ih = int(rng.integers(2, 400)), int(rng.integers(2, 300))
    rgba = _test_image(iw, ih, seed, alpha=bool(rng.integers(2)))
    path = str(tmp_path / "centre.png")
    _write_png(path, rgba)
    vs = dict(center_image_path=path, center_image_zoom=int(rng.choice([25, 60, 100, 100, 170, 300])),
              center_motion=int(rng.choice([0, 0, 30, 100])), radial_rotation_deg=float(rng.choice([0.0, 33.0])))
    if rng.random() < 0.5:
        vs.update(edge_waviness=int(rng.integers(1, 101)), feather_audio_enabled=True,
                  feather_audio_amount=int(rng.integers(0, 101)))
    cfg = RadialConfig.from_view_state(vs, 1.0)
    track = eng.upload_track(synth.pink(2500, 2, seed=40 + seed) * float(rng.choice([0.5, 2.0])), 48000, 60)
    pipe = FramePipeline(eng, track, "Spectrum - Radial", W, H, radial=cfg)
    an, sc = pipe.analyse()
    frames = _np(pipe.render_device())
    verts, env = eng.radial_vertices(pipe._radial_prep, an, sc, W, H, 0, an.n_frames)
    verts, env, pen = _np(verts), _np(env), _np(sc.pen_rgb)[:, :3]
    energy, phase = _np(sc.energy), _np(sc.phase)
    pm = R.premultiply(rgba)
    for f in range(an.n_frames):
        pl = R.center_image_placement(env[f], iw, ih, cfg.center_image_zoom, cfg.center_motion, float(energy[f]),
                                      float(phase[f]))
        want = R.render_radial_frame(verts[f], env[f], pen[f], W, H, image_pm=pm, placement=pl)
        assert np.array_equal(frames[f], want), (seed, (W, H), (iw, ih), vs, f, int(np.sum(frames[f] != want)))


def test_radial_unreadable_center_image_keeps_the_fill(eng, tmp_path):
    from aurora_amd import synth
    from aurora_amd.engine import RadialConfig
    from aurora_amd.pipeline import FramePipeline
    W, H = 320, 200
    bad = tmp_path / "broken.png"
    bad.write_bytes(b"not an image")
    track = eng.upload_track(synth.pink(2000, 2, seed=4), 48000, 60)
    a = _np(FramePipeline(eng, track, "Spectrum - Radial", W, H,
                          radial=RadialConfig.from_view_state(dict(center_image_path=str(bad)), 1.0)).render_device())
    b = _np(FramePipeline(eng, track, "Spectrum - Radial", W, H).render_device())
    assert np.array_equal(a, b)


@pytest.mark.parametrize("mode,off,dim,isize", [("fill", (0, 0), 0, (50, 40)), ("fit", (7, -5), 30, (50, 40)),
                                                ("stretch", (0, 0), 100, (33, 47)), ("center", (-11, 9), 55, (50, 40)),
                                                ("tile", (13, 6), 10, (50, 40)), ("fill", (3, 3), 0, (400, 300)),
                                                ("center", (0, 0), 0, (300, 200)), ("tile", (-7, -9), 0, (31, 29))])
def test_background_composition_bit_exact(eng, tmp_path, mode, off, dim, isize):
    pytest.importorskip("PIL")
    from aurora_amd.images import BackgroundConfig
    W, H = 160, 96
    rgba = _test_image(isize[0], isize[1], 9, alpha=(mode in ("fit", "tile")))
    path = str(tmp_path / "bg.png")
    _write_png(path, rgba)
    bg = BackgroundConfig(path, mode, off[0], off[1], dim)
    base = _np(eng.compose_background(bg, W, H))
    want = R.compose_background(R.premultiply(rgba), W, H, mode, off, dim)
    assert np.array_equal(base, want), (mode, int(np.sum(base != want)))


@pytest.mark.parametrize("seed", range(12))
def test_random_background_plans(eng, tmp_path, seed):
    """Random frame size, image size (smaller and larger than the frame, extreme aspect ratios), scale mode, offsets and dim:
    the composed base frame (plan + smooth scaling in all four axis combinations + source-over + dim) against the oracle."""
    pytest.importorskip("PIL")
    from aurora_amd.images import BackgroundConfig
    rng = np.random.default_rng(900 + seed)
    W, H = int(rng.integers(64, 400)), int(rng.integers(16, 260))
    iw, ih = int(rng.integers(1, 500)), int(rng.integers(1, 400))
    mode = ["fill", "fit", "stretch", "center", "tile"][seed % 5]
    off = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
    dim = int(rng.choice([0, 0, 15, 60, 100]))
    rgba = _test_image(iw, ih, seed, alpha=bool(rng.integers(2)))
    path = str(tmp_path / "bg.png")
    _write_png(path, rgba)
    base = _np(eng.compose_background(BackgroundConfig(path, mode, off[0], off[1], dim), W, H))
    want = R.compose_background(R.premultiply(rgba), W, H, mode, off, dim)
    assert np.array_equal(base, want), (seed, (W, H), (iw, ih), mode, off, dim, int(np.sum(base != want)))


@pytest.mark.parametrize("size", [(320, 192), (336, 190), (330, 190), (323, 191)])
@pytest.mark.parametrize("mode", ["Spectrum - Linear", "Waveform - Linear", "Particles", "Spectrum - Radial",
                                  "Waveform - Circular"])
def test_frames_over_a_background_image(eng, tmp_path, mode, size):
    """Every mode, widths of every fill path (multiple of 32 / 16 / odd): the frame is the oracle's frame painted on the
    composed background instead of the plain fill; a second pipeline without a background is unaffected."""
    pytest.importorskip("PIL")
    from aurora_amd import synth
    from aurora_amd.images import BackgroundConfig
    from aurora_amd.pipeline import FramePipeline
    from oracle import radial_ref as RR
    W, H = size
    rgba = _test_image(64, 48, 21)
    path = str(tmp_path / "bg.png")
    _write_png(path, rgba)
    bg = BackgroundConfig(path, "fill", 0, 0, 20)
    base = R.compose_background(R.premultiply(rgba), W, H, "fill", (0, 0), 20)
    track = eng.upload_track(synth.pink(2500, 2, seed=6), 48000, 60)
    pipe = FramePipeline(eng, track, mode, W, H, background=bg)
    an, sc = pipe.analyse()
    frames = _np(pipe.render_device())
    ls, energy, pen = _np(an.logspec), _np(sc.energy), _np(sc.pen_rgb)[:, :3]
    smp = _np(an.samples) if an.samples is not None else None
    for f in range(an.n_frames):
        if mode == "Spectrum - Radial":
            verts, env = eng.radial_vertices(pipe._radial_prep, an, sc, W, H, f, 1)
            want = R.render_radial_frame(_np(verts)[0], _np(env)[0], pen[f], W, H, base=base)
        elif mode == "Waveform - Circular":
            want = R.render_wave_circular_frame(RR.wave_circular_path(smp[f], float(energy[f]), W, H), pen[f], W, H,
                                                bg=base)
        else:
            want = R.render_frame(mode, smp[f] if smp is not None else None, ls[f], float(energy[f]), pen[f], W, H,
                                  bg=base)
        assert np.array_equal(frames[f], want), (mode, size, f, int(np.sum(frames[f] != want)))
    plain = _np(FramePipeline(eng, track, mode, W, H).render_device(first=0, count=1))[0]
    assert (plain[0, 0] == np.asarray(R.BG, np.uint8)).all()


def test_exporter_takes_image_keys(eng, tmp_path):
    """view_state keys of the reference (export/exporter.py:141-179) reach the renderer; a missing background file means no
    background, as `QImage(path).isNull()` does in the reference (rt_widget.py:508-509)."""
    pytest.importorskip("PIL")
    from aurora_amd import synth
    from aurora_amd.export.exporter import HipOffscreenRenderer
    rgba = _test_image(40, 30, 2)
    bgp, cip = str(tmp_path / "b.png"), str(tmp_path / "c.png")
    _write_png(bgp, rgba)
    _write_png(cip, rgba[::-1].copy())
    audio = synth.pink(2400, 2, seed=9)
    r = HipOffscreenRenderer(320, 192, "Spectrum - Radial", (0.2, 0.8, 1.0), 1.0, 60, (audio, 48000),
                             background_path=bgp, background_scale_mode="fit", background_dim_percent=25,
                             center_image_path=cip, center_motion=20)
    f0 = r.render_time(0.0)
    r2 = HipOffscreenRenderer(320, 192, "Spectrum - Radial", (0.2, 0.8, 1.0), 1.0, 60, (audio, 48000),
                              background_path=str(tmp_path / "missing.png"))
    g0 = r2.render_time(0.0)
    assert f0.shape == g0.shape == (192, 320, 3) and not np.array_equal(f0, g0)
    assert (g0[0, 0] == np.asarray(R.BG, np.uint8)).all() and not (f0[96, 0] == np.asarray(R.BG, np.uint8)).all()


def test_wavy_disc_at_4k_matches_oracle(eng):
    """The frame-dependent disc at 3840x2160 (box 2281 wide: the kernel's accumulators need more than 64 KB of LDS, the
    path that raises the kernel's dynamic-LDS limit), a few frames against the oracle's coverage and blend; and the
    frame-independent disc's rim list at the same size."""
    import torch
    from aurora_amd import synth
    from aurora_amd.engine import RadialConfig
    W, H = 3840, 2160
    track = eng.upload_track(synth.pink(2400, 2, seed=15), 48000, 60)
    an = eng.analyze(track)
    sc = eng.scan(an)
    for vs in (dict(edge_waviness=80, feather_audio_enabled=True, feather_audio_amount=100), {}):
        cfg = RadialConfig.from_view_state(vs, 1.0)
        prep = eng.radial_prepare(an, cfg)
        n = 2
        frames = torch.empty((n, H, W, 3), dtype=torch.uint8, device=eng.device)
        frames[:] = torch.tensor(R.BG, dtype=torch.uint8, device=eng.device)
        if cfg.disc_static:                                   # what pass 1 of the plane fill paints ahead of the rim
            full = eng._static_disc(prep, an, sc, W, H, 0)["full"]
            m = torch.from_numpy(R.unpack_mask(_np(full)[0].view(np.uint32), W)).to(eng.device)
            frames[:, m] = torch.tensor(R.DISC_RGB, dtype=torch.uint8, device=eng.device)
        eng.draw_disc(prep, an, sc, W, H, 0, n, frames)
        got = _np(frames)
        env = _np(eng._envelope(prep, an, sc, W, H, 0, n))
        for f in range(n):
            want = np.empty((H, W, 3), np.uint8)
            want[:] = np.asarray(R.BG, np.uint8)
            R.draw_center_disc(want, env[f], W, H)
            assert np.array_equal(got[f], want), (vs, f, int(np.sum(got[f] != want)))


@pytest.mark.parametrize("blend", ["normal", "add", "multiply"])
def test_fill_decoration_over_a_background_image(eng, tmp_path, blend):
    """`_apply_fill` (rt_widget.py:856-864) when the frame under it is a background image: the three composition modes
    per pixel (over the plain background the result is one colour, composed on the host)."""
    pytest.importorskip("PIL")
    from aurora_amd import synth
    from aurora_amd.engine import FillConfig
    from aurora_amd.images import BackgroundConfig
    from aurora_amd.pipeline import FramePipeline
    from oracle import radial_ref as RR
    W, H = 320, 192
    rgba = _test_image(64, 48, 33)
    path = str(tmp_path / "bg.png")
    _write_png(path, rgba)
    base = R.compose_background(R.premultiply(rgba), W, H, "stretch", (0, 0), 0)
    fc = FillConfig(enabled=True, rgba=(200, 90, 40, 150), blend=blend, threshold=0.0)
    track = eng.upload_track(synth.pink(2500, 2, seed=6), 48000, 60)
    pipe = FramePipeline(eng, track, "Waveform - Circular", W, H, fill=fc,
                         background=BackgroundConfig(path, "stretch", 0, 0, 0))
    an, sc = pipe.analyse()
    frames = _np(pipe.render_device())
    energy, pen, smp = _np(sc.energy), _np(sc.pen_rgb)[:, :3], _np(an.samples)
    r, g, b, a = fc.rgba
    for f in range(an.n_frames):
        verts = RR.wave_circular_path(smp[f], float(energy[f]), W, H)
        want = base.copy()
        inside = R.unpack_mask(R.polygon_mask(verts, W, H), W)
        d = want[inside].astype(np.uint32)
        s = R.byte_mul(np.asarray([r, g, b], np.uint32), a)
        if blend == "add":
            out = np.minimum(d + s, 255)
        elif blend == "multiply":
            out = R.byte_mul(d, s + 255 - a)
        else:
            out = s + R.byte_mul(d, 255 - a)
        want[inside] = out.astype(np.uint8)
        want[R.unpack_mask(R.stroke_mask_closed(verts, W, H), W)] = pen[f]
        assert np.array_equal(frames[f], want), (blend, f, int(np.sum(frames[f] != want)))
