"""ORACLE (test infrastructure, not product code): numpy restatement of the opt-in yuv420p / nv12 wire formats.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

The reference pipes rgb24 into FFmpeg (``export/exporter.py:429-434``) and asks FFmpeg for yuv420p output
(``:485-490``); the conversion itself lives in FFmpeg's swscale, which is neither in the reference tree nor installed
here: **parity unpinned**.  The rule is therefore stated in full (ITU-R BT.601 limited range, 8-bit integer form,
chroma from the rounded mean of each 2x2 block) and the HIP kernel (``csrc/wire.hip``) must match it bit for bit.
"""
from __future__ import annotations

import numpy as np


def rgb24_to_yuv420p(frame: np.ndarray) -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [H * W * 3 // 2]: Y plane [H, W], then Cb [H/2, W/2], then Cr [H/2, W/2]."""
    h, w, _ = frame.shape
    assert h % 2 == 0 and w % 2 == 0
    f = frame.astype(np.int32)
    r, g, b = f[..., 0], f[..., 1], f[..., 2]
    y = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16
    m = (f.reshape(h // 2, 2, w // 2, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    mr, mg, mb = m[..., 0], m[..., 1], m[..., 2]
    cb = ((-38 * mr - 74 * mg + 112 * mb + 128) >> 8) + 128
    cr = ((112 * mr - 94 * mg - 18 * mb + 128) >> 8) + 128
    return np.concatenate([y.reshape(-1), cb.reshape(-1), cr.reshape(-1)]).astype(np.uint8)


def rgb24_to_nv12(frame: np.ndarray) -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [H * W * 3 // 2]: Y plane [H, W], then one plane [H/2, W] of interleaved Cb Cr pairs
    (the layout the reference asks FFmpeg for on the VAAPI path, ``export/exporter.py:464-466``).  Same values as
    ``rgb24_to_yuv420p``, chroma interleaved instead of planar."""
    h, w, _ = frame.shape
    p = rgb24_to_yuv420p(frame)
    y, cb, cr = p[:h * w], p[h * w:h * w + h * w // 4], p[h * w + h * w // 4:]
    return np.concatenate([y, np.stack([cb, cr], axis=1).reshape(-1)])
