#!/usr/bin/env python3
"""Micro-benchmark of the fused STFT kernel alone (GPU box): ms per launch, frames/s, algorithmic GB/s.

    python tools/stft_bench.py [--seconds 3600] [--reps 10]

Parity of the kernel is the tests' business (tests/test_gpu_parity.py, tests/test_gpu_e2e_parity.py): nothing outside
tests/, smoke() and bench.py's cpu_baseline touches oracle/.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3600.0)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from aurora_amd import synth
    from aurora_amd.engine import get_engine

    eng = get_engine(0)
    sr, fps = 48000, 60
    audio = synth.pink_device(int(args.seconds * sr), 2, 1234, eng.device)
    track = eng.make_track(audio, sr, fps)
    F = track.total_frames
    for _ in range(2):
        an = eng.analyze(track)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
    evs[0].record()
    for i in range(args.reps):
        an = eng.analyze(track)
        evs[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.reps))
    ms = sum(per) / len(per)
    bytes_per_frame = 4 * 2 * min(track.hop, 1024) + 4 * 1025 + 8
    gbs = F * bytes_per_frame / (ms * 1e-3) / 1e9
    print(f"stft: {F} frames  {ms:.4f} ms/launch  {F / ms * 1e3 / 1e6:.1f} Mframes/s  {gbs:.1f} GB/s "
          f"({gbs / 8000 * 100:.1f}% of 8 TB/s)  median {per[len(per) // 2]:.4f}  min {per[0]:.4f}")


if __name__ == "__main__":
    main()
