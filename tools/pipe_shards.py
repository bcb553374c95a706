#!/usr/bin/env python3
"""One GPU, K time shards, each with its own pipe and its own reading child (the shape of the multi-GPU driver's
per-shard FFmpeg, aurora_amd/export/sharded.py, collapsed onto one card): how does the rate into the pipes scale with the
number of readers?  Every shard is a fresh process (engine, pinned ring, pipe sink, tools/pipe_drain.py as the reader).

    python tools/pipe_shards.py --shards 1 2 4 [--frames 960] [--threads 4]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shard_main(rank, world, frames, wire):
    import torch
    from aurora_amd import sharding, synth
    from aurora_amd.engine import get_engine
    from aurora_amd.export.exporter import _PipeSink, _grow_pipe
    from aurora_amd.pipeline import FramePipeline
    eng = get_engine(0)
    sr, fps, W, H = 48000, 60, 3840, 2160
    audio = synth.pink_device(int(frames / fps * sr), 2, 1234, eng.device)
    track = eng.make_track(audio, sr, fps)
    a, b = sharding.frame_range(track.total_frames, world, rank)
    pipe = FramePipeline(eng, track, "Spectrum - Linear", W, H, frame_begin=a, frame_end=b, batch_frames=32, slots=3,
                         wire_format=wire).prepare()
    pipe.analyse()
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "pipe_drain.py")], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, bufsize=10 ** 7)
    _grow_pipe(child.stdin)
    sink = _PipeSink(child.stdin)
    torch.cuda.synchronize()
    print("READY", flush=True)
    sys.stdin.readline()                      # all shards start together
    t0 = time.perf_counter()
    n = pipe.run(sink)
    child.stdin.close()
    rep = json.loads(child.stdout.read())
    child.wait()
    dt = time.perf_counter() - t0
    sink.close()
    print(json.dumps({"rank": rank, "frames": n, "bytes": rep["bytes"], "seconds": dt}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--frames", type=int, default=960)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--wire", default="rgb24")
    ap.add_argument("--shard", type=int, nargs=2, default=None)
    args = ap.parse_args()
    if args.shard is not None:
        return shard_main(args.shard[0], args.shard[1], args.frames, args.wire)
    for K in args.shards:
        env = dict(os.environ, AUR_PIPE_COPY_THREADS=str(args.threads))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--shard", str(r), str(K), "--frames",
                                   str(args.frames), "--wire", args.wire], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  text=True, env=env) for r in range(K)]
        for p in procs:
            assert p.stdout.readline().strip() == "READY"
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        reps = [json.loads(p.stdout.readline()) for p in procs]
        wall = time.perf_counter() - t0
        for p in procs:
            p.wait()
        frames = sum(r["frames"] for r in reps)
        nbytes = sum(r["bytes"] for r in reps)
        print(json.dumps({"shards": K, "copy_threads": args.threads, "wire": args.wire, "frames": frames,
                          "fps": round(frames / wall, 1), "GBps": round(nbytes / wall / 1e9, 2),
                          "slowest_shard_s": round(max(r["seconds"] for r in reps), 2)}), flush=True)


if __name__ == "__main__":
    main()
