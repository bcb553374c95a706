"""Raw spectral flux of the GPU kernel against the flux recomputed from its own log-spectra, frame by frame (GPU box).
A chunk whose halo frame hands wrong previous magnitudes on shows up as one frame in every frames-per-wave.  Test infrastructure:
reads the fixtures under tests/golden.

    python tools/stft_flux_check.py      (AUR_STFT_FPW=n moves the chunk boundaries)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.conftest import load_golden
from aurora_amd.engine import get_engine
from oracle import analysis_ref as A
eng = get_engine(0)
for name in ("pink_48k_60", "pink_44k1_60"):
    g = load_golden("analysis_" + name)
    audio, sr, fps = g["audio"], int(g["sample_rate"]), int(g["fps"])
    track = eng.upload_track(audio, sr, fps)
    an = eng.analyze(track, want_samples=True, want_bandmax=True)
    fs = an.flux_sum.cpu().numpy() if hasattr(an, "flux_sum") else None
    # reference raw flux from the fixture's logspec: mag = expm1(logspec)
    mag = np.expm1(g["logspec"].astype(np.float64))
    prev = np.vstack([np.zeros((1, mag.shape[1])), mag[:-1]])
    ref = np.maximum(mag - prev, 0).sum(axis=1)
    print(name, "hop", track.hop, "frames", len(ref))
    if fs is not None:
        d = (fs - ref) / ref
        print(" rel diff of raw flux, first 24 frames:", np.array2string(d[:24], precision=4, max_line_width=200))
        bad = np.nonzero(np.abs(d) > 1e-4)[0]
        print(" frames off by > 1e-4:", bad[:40], "count", len(bad))
