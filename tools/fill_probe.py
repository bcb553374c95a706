#!/usr/bin/env python3
"""Where does the fill kernel stand against a plain fill of the same slot?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aurora_amd import _lib
from aurora_amd.engine import get_engine

eng = get_engine(0)
B, H, W, K = 128, 2160, 3840, 2
frames = [eng.empty((B, H, W, 3), torch.uint8) for _ in range(2)]
pen = torch.zeros((B, 4), dtype=torch.uint8, device="cuda")
pen[:, 0] = 200
empty = torch.zeros((B, K, W), dtype=torch.int32, device="cuda")
bars = empty.clone()
for i in range(80):
    x0 = i * 48
    v = (300 + 13 * i) | (2156 << 16)
    bars[:, 0, x0] = v
    if x0 > 0:
        bars[:, 1, x0 - 1] = v


def timeit(name, fn, n=20):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:34s} {ms:.4f} ms  {B * H * W * 3 / ms / 1e6:.0f} GB/s")


timeit("torch fill_ (int32 view)", lambda i: frames[i % 2].view(-1).view(torch.int32).fill_(0x0d0d120d))
timeit("fill kernel, empty spans", lambda i: eng.fill(empty, K, pen, W, H, out=frames[i % 2]))
timeit("fill kernel, 80 bars", lambda i: eng.fill(bars, K, pen, W, H, out=frames[i % 2]))
