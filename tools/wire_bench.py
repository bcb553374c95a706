#!/usr/bin/env python3
"""Micro-benchmark of the rgb24 -> yuv420p wire kernel (GPU box): ms per launch and HBM GB/s (4.5 B per pixel)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from aurora_amd import _lib
    from aurora_amd.engine import get_engine
    eng = get_engine(0)
    for (W, H, n) in ((3840, 2160, 64), (1920, 1080, 256)):
        rgb = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device=eng.device)
        yuv = torch.empty((n, W * H * 3 // 2), dtype=torch.uint8, device=eng.device)

        def run():
            _lib.check(eng.lib.aur_rgb24_to_yuv420p(eng.ctx.handle, C.c_void_p(rgb.data_ptr()), n, W, H,
                                                    C.c_void_p(yuv.data_ptr()), eng.stream()), "aur_rgb24_to_yuv420p")
        for _ in range(2):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gbs = n * W * H * 4.5 / (ms * 1e-3) / 1e9
        print(f"yuv420p {W}x{H} x{n}: {ms:.3f} ms/launch  {n / ms * 1e3:.0f} frames/s  {gbs:.0f} GB/s "
              f"({gbs / 80:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
