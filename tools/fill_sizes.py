#!/usr/bin/env python3
"""Linear fill kernel alone at the frame sizes of BASELINE.json's configurations: empty spans, the spectrum bars and the
particles of a pink-noise track, per batch size.  (GPU box; events around back-to-back fill launches, summaries ready.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aurora_amd import _lib, synth
from aurora_amd.engine import get_engine

eng = get_engine(0)
sr, fps = 48000, 60
track = eng.make_track(synth.pink_device(sr * 8, 2, 1234, eng.device), sr, fps)
an = eng.analyze(track)
sc = eng.scan(an)


def timeit(fn, n=40):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (W, H) in [(1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)]:
    for B in (64, 128):
        frames = [eng.empty((B, H, W, 3), torch.uint8) for _ in range(2)]
        row = []
        for kind in ("empty", "spectrum_linear", "particles"):
            K = 2
            if kind == "empty":
                spans = torch.zeros((B, K, W), dtype=torch.int32, device=eng.device)
            else:
                spans, K = eng.spans(kind, an, sc, W, H, 0, B)
            work = eng.summarize(spans, K, W, H)
            pen = sc.pen_rgb[:B]
            ms = timeit(lambda i: eng.fill(spans, K, pen, W, H, out=frames[i % 2], work=work, work_ready=True))
            row.append(f"{kind} {ms:.4f} ms {B * H * W * 3 / ms / 1e6:6.0f} GB/s")
        print(f"{W}x{H} B={B}: " + " | ".join(row), flush=True)
        del frames
