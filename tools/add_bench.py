"""Times aur_add_planes (glow layers) alone on one GPU: real ring planes, all-zero planes, and real planes with the
per-layer stroke time next to it.   python tools/add_bench.py [--frames 128] [--width 3840 --height 2160]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import torch
    from aurora_amd import _lib
    from aurora_amd.engine import GlowConfig, _ptr, get_engine
    eng = get_engine(0)
    F, W, H = a.frames, a.width, a.height
    W32 = (W + 31) // 32
    nv = 722
    ang = 2 * np.pi * np.arange(nv) / nv
    rad = 0.22 * min(W, H) + 0.2 * min(W, H) * (0.5 + 0.5 * np.sin(7 * ang))
    verts = np.stack([W / 2 + np.cos(ang) * rad, H / 2 + np.sin(ang) * rad], axis=1)
    verts = torch.from_numpy(np.broadcast_to(verts, (F, nv, 2)).copy()).to(eng.device)
    glow = GlowConfig(enabled=True)
    table = glow.layer_table()
    K = len(table)
    planes = torch.zeros((K, F, H, W32), dtype=torch.int32, device=eng.device)
    stroke = torch.zeros((F, H, W32), dtype=torch.int32, device=eng.device)
    h, st = eng.ctx.handle, eng.stream()

    def timed(fn, reps=a.reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for k, (width, _) in enumerate(table):
        ms = timed(lambda: _lib.check(eng.lib.aur_mask_stroke_path(h, _ptr(verts), nv, F, W, H, width / 2.0, 0.0, 0.0,
                                                                   1, 1, _ptr(planes[k]), st), "stroke"), 3)
        print(f"stroke plane width {width}: {ms:.3f} ms per {F} frames")
    _lib.check(eng.lib.aur_mask_stroke_closed(h, _ptr(verts), nv, F, W, H, _ptr(stroke), st), "stroke")
    covered = int(torch.count_nonzero(planes[K - 1]).item())
    pm = torch.from_numpy(glow.layer_premultiplied(np.full(F, 0.5))).to(eng.device)
    zero_pm = torch.zeros_like(pm)
    frames = torch.full((F, H, W, 3), 13, dtype=torch.uint8, device=eng.device)
    zeros = torch.zeros_like(planes)

    def add(pl, colours, ea=None):
        _lib.check(eng.lib.aur_add_planes(h, _ptr(pl), K, _ptr(colours), _ptr(ea), None, 0, F, W, H, _ptr(frames), st),
                   "aur_add_planes")

    gb = planes.numel() * 4 / 1e9
    for name, fn in (("ring planes", lambda: add(planes, pm)), ("ring planes + stroke exclusion", lambda: add(planes, pm, stroke)),
                     ("ring planes, zero colours", lambda: add(planes, zero_pm)), ("all-zero planes", lambda: add(zeros, pm))):
        ms = timed(fn)
        print(f"aur_add_planes {name}: {ms:.3f} ms per {F} frames x {K} layers ({gb / ms:.2f} TB/s of plane reads; "
              f"{covered} non-zero words in the widest plane)")


if __name__ == "__main__":
    main()
