"""Synthetic audio tracks for tests and benchmarks.

Two generators, both float32, peak 0.8, shape ``[n_samples, channels]`` interleaved
(the layout ``soundfile.SoundFile.read(dtype='float32')`` hands the reference exporter,
``export/exporter.py:232``):

* ``sweep``: linear sine chirp 20 Hz -> 20 kHz over the track, right channel is the left
  channel delayed by half a track (circular).
* ``pink``: white gaussian noise shaped by 1/sqrt(k) in the frequency domain, one seed
  per channel.

Host (numpy) versions are used for fixtures and small tests; ``*_device`` versions build the
long benchmark tracks directly in HBM with torch (tensor plumbing only).
"""
from __future__ import annotations

import numpy as np

PEAK = 0.8


def sweep(n_samples: int, sample_rate: int, channels: int = 2,
          f0: float = 20.0, f1: float = 20000.0) -> np.ndarray:
    n = int(n_samples)
    t = np.arange(n, dtype=np.float64) / float(sample_rate)
    T = n / float(sample_rate)
    phase = 2.0 * np.pi * (f0 * t + (f1 - f0) / (2.0 * T) * t * t)
    left = PEAK * np.sin(phase)
    out = np.empty((n, channels), dtype=np.float32)
    out[:, 0] = left.astype(np.float32)
    for c in range(1, channels):
        out[:, c] = np.roll(left, -(n // 2) * c).astype(np.float32)
    return out


def pink(n_samples: int, channels: int = 2, seed: int = 1234) -> np.ndarray:
    n = int(n_samples)
    out = np.empty((n, channels), dtype=np.float32)
    for c in range(channels):
        rng = np.random.default_rng(seed + 10_000 * c)
        white = rng.standard_normal(n)
        spec = np.fft.rfft(white)
        k = np.arange(spec.shape[0], dtype=np.float64)
        spec /= np.sqrt(np.maximum(k, 1.0))
        y = np.fft.irfft(spec, n=n)
        y *= PEAK / max(1e-12, float(np.max(np.abs(y))))
        out[:, c] = y.astype(np.float32)
    return out


def make(kind: str, n_samples: int, sample_rate: int, channels: int = 2, seed: int = 1234) -> np.ndarray:
    if kind == "sweep":
        return sweep(n_samples, sample_rate, channels)
    if kind == "pink":
        return pink(n_samples, channels, seed)
    raise ValueError(f"unknown synthetic audio kind: {kind!r}")


def pink_device(n_samples: int, channels: int, seed: int, device):
    """Pink-noise track generated in HBM (benchmark input; torch used as plumbing only)."""
    import torch

    n = int(n_samples)
    out = torch.empty((n, channels), dtype=torch.float32, device=device)
    for c in range(channels):
        g = torch.Generator(device=device)
        g.manual_seed(int(seed) + 10_000 * c)
        white = torch.randn(n, generator=g, device=device, dtype=torch.float32)
        spec = torch.fft.rfft(white)
        del white
        k = torch.arange(spec.shape[0], device=device, dtype=torch.float32)
        spec /= torch.sqrt(torch.clamp(k, min=1.0))
        del k
        y = torch.fft.irfft(spec, n=n)
        del spec
        y *= PEAK / float(torch.max(torch.abs(y)).clamp(min=1e-12))
        out[:, c] = y
        del y
    return out


def sweep_device(n_samples: int, sample_rate: int, channels: int, device,
                 f0: float = 20.0, f1: float = 20000.0):
    import torch

    n = int(n_samples)
    t = torch.arange(n, device=device, dtype=torch.float64) / float(sample_rate)
    T = n / float(sample_rate)
    left = (PEAK * torch.sin(2.0 * torch.pi * (f0 * t + (f1 - f0) / (2.0 * T) * t * t))).to(torch.float32)
    del t
    out = torch.empty((n, channels), dtype=torch.float32, device=device)
    out[:, 0] = left
    for c in range(1, channels):
        out[:, c] = torch.roll(left, -(n // 2) * c)
    return out
