"""End-to-end frame parity at FULL size: GPU audio -> GPU spectrum -> GPU frames against the oracle rasteriser fed the
ORACLE'S OWN spectrum (not the GPU's), for BASELINE.json configs[1], [2], [3] and the 4K headline mode.

What can differ: the displayed bins go through ``int(v * (h - 10) * ...)`` (rt_widget.py:986) and
``int(h / 2 + (v - 0.5) * h * 0.6)`` (:997), so a last-ulp difference in ``log1p|X|`` moves a bar top or a particle by one
row in a small fraction of frames.  The test counts those frames over >= 2000 frames sampled evenly from the 4 minute
track, prints the rate (recorded in DESIGN.md 2) and bounds it.  The waveform mode depends on samples and energy only,
which are bit-exact, so its frames must be identical.
"""
import multiprocessing as mp
import os
from multiprocessing import shared_memory

import numpy as np
import pytest

from oracle import analysis_ref as A

pytestmark = pytest.mark.gpu

N_SAMPLED = 2000
BATCH = 16


@pytest.fixture(scope="module")
def eng():
    from aurora_amd.engine import get_engine
    return get_engine(0)


@pytest.fixture(scope="module")
def track4min(eng):
    from aurora_amd import synth
    sr, fps = 48000, 60
    audio = synth.pink_device(sr * 240, 2, 1234, eng.device)
    track = eng.make_track(audio, sr, fps)
    an = eng.analyze(track, want_samples=True)
    sc = eng.scan(an)
    ref = A.analyze_track(audio.cpu().numpy(), sr, fps)            # the oracle's own analysis of the same samples
    pen, _ = A.amp_to_color_track(ref["energy"])
    return track, an, sc, ref, pen


@pytest.fixture(scope="module")
def pool():
    import _frame_compare_worker  # noqa: F401  (tests/ is on sys.path; spawned workers import it by name)
    ctx = mp.get_context("spawn")                                  # never fork a process that has initialised the GPU
    with ctx.Pool(min(8, max(2, (os.cpu_count() or 2) // 2))) as p:
        yield p


# measured on MI355X: frames with any differing pixel / frames compared
MAX_RATE = {"Spectrum - Linear": 0.004, "Particles": 0.004, "Waveform - Linear": 0.0}


@pytest.mark.parametrize("mode,size", [("Spectrum - Linear", (1920, 1080)), ("Waveform - Linear", (3840, 2160)),
                                       ("Particles", (1920, 1080)), ("Spectrum - Linear", (3840, 2160))])
def test_frames_vs_oracle_own_spectrum(eng, track4min, pool, mode, size):
    import torch
    from _frame_compare_worker import compare_frame
    track, an, sc, ref, pen = track4min
    W, H = size
    F = track.total_frames
    picks = np.unique(np.linspace(0, F - 1, N_SAMPLED).astype(np.int64))
    fb = W * H * 3
    shms = [shared_memory.SharedMemory(create=True, size=BATCH * fb) for _ in range(2)]
    try:
        views = [np.ndarray((BATCH, H, W, 3), dtype=np.uint8, buffer=s.buf) for s in shms]
        dev = eng.empty((BATCH, H, W, 3), torch.uint8)
        pending, results = None, []
        for b0 in range(0, len(picks), BATCH):
            group = picks[b0:b0 + BATCH]
            half = (b0 // BATCH) % 2
            for j, f in enumerate(group):
                eng.render(mode, an, sc, W, H, int(f), 1, out=dev[j:j + 1])
            if pending is not None and half == pending[1]:
                raise AssertionError("double buffering out of step")
            host = dev[:len(group)].cpu().numpy()
            # the half being overwritten was compared two batches ago (its map_async result has been collected)
            views[half][:len(group)] = host
            tasks = [(shms[half].name, j * fb, mode, ref["samples"][f], ref["logspec"][f], float(ref["energy"][f]),
                      pen[f], W, H) for j, f in enumerate(group)]
            if pending is not None:
                results.extend(pending[0].get(timeout=300))
            pending = (pool.map_async(compare_frame, tasks), half)
        results.extend(pending[0].get(timeout=300))
        del views
    finally:
        for s in shms:
            s.close()
            s.unlink()
    assert len(results) == len(picks)
    ndiff = np.array([r[0] for r in results])
    nbad = np.array([r[1] for r in results])
    ps = np.array([r[2] for r in results])
    flipped = int(np.count_nonzero(ndiff))
    rate = flipped / len(picks)
    print(f"\n{mode} {W}x{H}: {flipped} of {len(picks)} sampled frames differ from the oracle's own export "
          f"(rate {rate:.5f}); worst frame: {int(ndiff.max())} pixels differ, {int(nbad.max())} by more than 1 LSB, "
          f"min PSNR {ps.min():.1f} dB")
    assert ps.min() >= 50.0
    # a flip moves one bar top (2 px wide) or one particle (2x2 px, leaves 2 and enters 2) by a row
    assert nbad.max() <= 8
    assert rate <= MAX_RATE[mode], rate


def test_config0_sweep_every_frame(eng, pool):
    """BASELINE.json configs[0] in full: the 30 s 44.1 kHz stereo sine sweep, "Spectrum - Linear" 1280x720@30, every one
    of its 900 frames against the oracle's export of the oracle's own analysis.  The sweep is the hard input for the
    spectrum: most bins of a frame are leakage, decades under the tone."""
    import torch
    from _frame_compare_worker import compare_frame
    from aurora_amd import synth
    sr, fps, W, H = 44100, 30, 1280, 720
    audio = synth.sweep(sr * 30, sr, 2)
    track = eng.upload_track(audio, sr, fps)
    an = eng.analyze(track, want_samples=True)
    sc = eng.scan(an)
    ref = A.analyze_track(audio, sr, fps)
    pen, _ = A.amp_to_color_track(ref["energy"])
    F = track.total_frames
    assert F == 900 and np.array_equal(sc.pen_rgb.cpu().numpy()[:, :3], pen)
    got = an.logspec.cpu().numpy()
    assert np.allclose(got, ref["logspec"], rtol=1e-4, atol=2e-6)           # the north star's spectrum tolerance
    fb = W * H * 3
    n_batch = 60
    shm = shared_memory.SharedMemory(create=True, size=n_batch * fb)
    try:
        view = np.ndarray((n_batch, H, W, 3), dtype=np.uint8, buffer=shm.buf)
        results = []
        for b0 in range(0, F, n_batch):
            n = min(n_batch, F - b0)
            view[:n] = eng.render("Spectrum - Linear", an, sc, W, H, b0, n).cpu().numpy()
            tasks = [(shm.name, j * fb, "Spectrum - Linear", ref["samples"][b0 + j], ref["logspec"][b0 + j],
                      float(ref["energy"][b0 + j]), pen[b0 + j], W, H) for j in range(n)]
            results.extend(pool.map(compare_frame, tasks))
        del view
    finally:
        shm.close()
        shm.unlink()
    ndiff = np.array([r[0] for r in results])
    ps = np.array([r[2] for r in results])
    flipped = int(np.count_nonzero(ndiff))
    print(f"\nconfigs[0] sweep 1280x720@30: {flipped} of {F} frames differ from the oracle's own export; worst frame "
          f"{int(ndiff.max())} pixels, min PSNR {ps.min():.1f} dB")
    assert ps.min() >= 50.0 and flipped <= 4 and ndiff.max() <= 8


@pytest.mark.parametrize("mode,size,n_sampled", [("Spectrum - Radial", (1920, 1080), 480),
                                                 ("Spectrum - Radial", (3840, 2160), 240),
                                                 ("Waveform - Circular", (3840, 2160), 480)])
def test_closed_path_frames_vs_oracle_own_chain(eng, track4min, pool, mode, size, n_sampled):
    """The closed-path modes end to end: GPU audio -> GPU spectrum -> smoothing chain, `_radial_prev` EMA, vertices,
    stroke, antialiased disc on the GPU, against the oracle's own chain from the oracle's own spectrum (radial_ref over
    every frame of the track in order, then the oracle rasteriser on the sampled frames).  "Waveform - Circular" depends
    on samples and energy only (bit-exact), so its frames must be identical; the radial path inherits the last-ulp
    differences of log1p|X| through normalisation and smoothing, so an edge pixel may flip in a few frames: the rate is
    printed and bounded."""
    import torch
    from _frame_compare_worker import compare_closed_frame
    from aurora_amd.engine import RadialConfig
    from oracle import radial_ref as RR
    track, an, sc, ref, pen = track4min
    W, H = size
    F = track.total_frames
    picks = np.unique(np.linspace(0, F - 1, n_sampled).astype(np.int64))
    wanted = set(int(f) for f in picks)
    radial = mode == "Spectrum - Radial"
    paths, envs = {}, {}
    if radial:
        phase = A.phase_track(ref["energy"])
        tr = RR.RadialTrack()
        for f in range(F):                                         # the EMA is sequential: every frame, in order
            if f in wanted:
                paths[f] = tr.path(ref["logspec"][f], float(ref["energy"][f]), W, H)
                envs[f] = RR.envelope_path(ref["logspec"][f], float(ref["energy"][f]), float(phase[f]), W, H)
            else:
                tr.values(ref["logspec"][f])
        prep = eng.radial_prepare(an, RadialConfig())
    else:
        for f in wanted:
            paths[f] = RR.wave_circular_path(ref["samples"][f], float(ref["energy"][f]), W, H)
            envs[f] = None
    fb = W * H * 3
    shms = [shared_memory.SharedMemory(create=True, size=BATCH * fb) for _ in range(2)]
    try:
        views = [np.ndarray((BATCH, H, W, 3), dtype=np.uint8, buffer=s.buf) for s in shms]
        dev = eng.empty((BATCH, H, W, 3), torch.uint8)
        pending, results = None, []
        for b0 in range(0, len(picks), BATCH):
            group = [int(f) for f in picks[b0:b0 + BATCH]]
            half = (b0 // BATCH) % 2
            for j, f in enumerate(group):
                if radial:
                    eng.render_radial(prep, an, sc, W, H, f, 1, out=dev[j:j + 1])
                else:
                    eng.render_wave_circular(an, sc, W, H, f, 1, out=dev[j:j + 1])
            host = dev[:len(group)].cpu().numpy()
            if pending is not None:
                results.extend(pending[0].get(timeout=600))       # the half written next was compared by this batch
            views[half][:len(group)] = host
            tasks = [(shms[half].name, j * fb, mode, paths[f], envs[f], pen[f], W, H) for j, f in enumerate(group)]
            pending = (pool.map_async(compare_closed_frame, tasks), half)
        results.extend(pending[0].get(timeout=600))
        del views
    finally:
        for s in shms:
            s.close()
            s.unlink()
    assert len(results) == len(picks)
    ndiff = np.array([r[0] for r in results])
    nbad = np.array([r[1] for r in results])
    ps = np.array([r[2] for r in results])
    flipped = int(np.count_nonzero(ndiff))
    rate = flipped / len(picks)
    print(f"\n{mode} {W}x{H}: {flipped} of {len(picks)} sampled frames differ from the oracle's own export "
          f"(rate {rate:.5f}); worst frame: {int(ndiff.max())} pixels differ, {int(nbad.max())} by more than 1 LSB, "
          f"min PSNR {ps.min():.1f} dB")
    if radial:
        # measured on MI355X: 1 of 480 frames at 1080p, 3 of 240 at 4K, one pixel each
        assert ps.min() >= 50.0 and ndiff.max() <= 2 and flipped <= (3 if H == 1080 else 6), (flipped, int(ndiff.max()))
    else:
        assert flipped == 0


def test_config3_union_composite_at_its_own_size(eng, track4min, pool):
    """BASELINE configs[3] ("particle mode + audio-reactive image feather", 1920x1080@60) at ITS size with the settings
    `bench.py --mode "Particles + Feather"` times: particle frames with the audio-reactive disc envelope and the six
    feather rings composited from planes of the feather box (`feather_blend(..., with_disc=True)`; the box depends on
    min(W, H), so 1080p is its own geometry).  48 frames sampled from the 4-minute track, two bars:
    (a) against the oracle composite of the GPU's own envelope / ring vertices and spectrum: bit-exact;
    (b) against the oracle's own chain (its spectrum, `envelope_path`, `feather_ring_paths`, rt_widget.py:1007-1051,
        :1098-1117): a vertex that differs in the last place (device sin / the spectrum's last ulp) may flip an edge pixel;
        the count is printed and bounded."""
    import torch
    from _frame_compare_worker import compare_union_frame
    from aurora_amd.engine import RadialConfig
    from oracle import radial_ref as RR
    track, an, sc, ref, pen = track4min
    W, H = 1920, 1080
    F = track.total_frames
    picks = [int(f) for f in np.unique(np.linspace(0, F - 1, 48).astype(np.int64))]
    cfg = RadialConfig(feather_enabled=True, feather_audio_enabled=True, feather_audio_amount=60, edge_waviness=40)
    params = RR.RadialParams(feather_enabled=True, feather_audio_enabled=True, feather_audio_amount=60, edge_waviness=40)
    prep = eng.radial_prepare(an, cfg)
    phase = A.phase_track(ref["energy"])
    alphas = [max(10, 70 - 10 * k) for k in range(1, 7)]
    fb = W * H * 3
    shm = shared_memory.SharedMemory(create=True, size=len(picks) * fb)
    try:
        view = np.ndarray((len(picks), H, W, 3), dtype=np.uint8, buffer=shm.buf)
        own, chain = [], []
        ls_gpu, en_gpu, pen_gpu = an.logspec, sc.energy, sc.pen_rgb
        for j, f in enumerate(picks):
            frame = eng.render("Particles", an, sc, W, H, f, 1)
            eng.feather_blend(prep, an, sc, W, H, f, 1, frame, with_disc=True)
            view[j] = frame[0].cpu().numpy()
            env = eng._envelope(prep, an, sc, W, H, f, 1)[0].cpu().numpy()
            rings = eng.feather_ring_vertices(prep, an, sc, W, H, f, 1)[0].cpu().numpy()
            own.append((shm.name, j * fb, ls_gpu[f].cpu().numpy(), float(en_gpu[f].item()),
                        pen_gpu[f].cpu().numpy()[:3], env, list(rings), alphas, W, H))
            e = float(ref["energy"][f])
            want_env = RR.envelope_path(ref["logspec"][f], e, float(phase[f]), W, H, params)
            want_rings = RR.feather_ring_paths(ref["logspec"][f], e, float(phase[f]), W, H, params)
            assert [a for _, a in want_rings] == alphas
            assert np.max(np.abs(env - want_env)) <= 2e-3 and max(np.max(np.abs(r - w[0])) for r, w in zip(rings, want_rings)) <= 2e-3
            chain.append((shm.name, j * fb, ref["logspec"][f], e, pen[f], want_env, [v for v, _ in want_rings], alphas, W, H))
        bx, by, bw, bh = eng.feather_box(cfg, W, H)
        assert (bw, bh) != (W, H) and bx % 32 == 0          # planes of the box, not of the frame
        res_own = pool.map(compare_union_frame, own)
        res_chain = pool.map(compare_union_frame, chain)
        changed = int(np.sum(np.any(view[0] != eng.render("Particles", an, sc, W, H, picks[0], 1)[0].cpu().numpy(), axis=2)))
        del view
    finally:
        shm.close()
        shm.unlink()
    assert changed > 50_000                                 # the disc and its rings are in the frame
    bad_own = [(f, r[0]) for f, r in zip(picks, res_own) if r[0]]
    assert not bad_own, bad_own
    ndiff = np.array([r[0] for r in res_chain])
    ps = np.array([r[2] for r in res_chain])
    print(f"\nconfigs[3] union composite 1920x1080: GPU == oracle on its own vertices for all {len(picks)} frames; against "
          f"the oracle's own chain {int(np.count_nonzero(ndiff))} frames differ, worst {int(ndiff.max())} pixels, "
          f"min PSNR {ps.min():.1f} dB")
    assert ps.min() >= 50.0 and ndiff.max() <= 4          # measured: 0 of 48 frames differ


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_displayed_bins_follow_the_reference_chain(eng, track4min):
    """log1p|X| of the bins the export draws (spectrum[:80], rt_widget.py:935): the magnitude is numpy's own float32
    complex-abs formula on the complex64-rounded spectrum (bit-exact), the log1p is correctly rounded where numpy's SVML
    routine is within 1-2 ulp of that; measured: >= 90 % of the values bit-equal to the oracle, none off by more than 2 ulp."""
    track, an, sc, ref, pen = track4min
    got = an.logspec[:, :80].cpu().numpy()
    want = ref["logspec"][:, :80]
    u = _ulps(got, want)
    eq = float(np.mean(u == 0))
    print(f"\ndisplayed bins over {got.shape[0]} frames: {eq * 100:.2f} % bit-equal to the oracle, max {int(u.max())} ulp; "
          f"all bins: max rel err {float(np.max(np.abs(an.logspec.cpu().numpy().astype(np.float64) - ref['logspec']) / np.maximum(ref['logspec'], 1e-30))):.3e}")
    assert eq >= 0.90 and u.max() <= 2
    # the magnitudes themselves: expm1 of the correctly rounded log1p is not the magnitude, so check them through the
    # streaming entry point, which returns prev_mag = |X| (Analyzer.prev_mag)
    from aurora_amd.audio.analysis import Analyzer
    rng = np.random.default_rng(3)
    a, r = Analyzer(48000, 2048), A.AnalyzerRef(48000, 2048)
    for _ in range(6):
        x = (rng.standard_normal(1024) * 0.3).astype(np.float32)
        a.compute(x)
        r.compute(x)
        assert np.array_equal(a.prev_mag[:128], r.prev_mag[:128])
