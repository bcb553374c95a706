#!/usr/bin/env python3
"""Benchmark of the MI355X export hot path (BASELINE.json metric: exported frames/s, 4K60 spectrum mode; STFT GB/s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (per GPU, per step): one unit of BASELINE.json configs[4] -- a 1 h, 48 kHz, stereo synthetic pink-noise
track exported in "Spectrum - Linear" mode at 3840x2160@60 = 216 000 frames.  A step runs the whole device path
over that track: fused STFT analysis (1 launch) -> smoothing scan -> column spans -> RGB24 fill of every frame
into a ring of device slots (frames stay in HBM; the PCIe-inclusive rate is measured separately and reported as
`pcie_inclusive_fps`, never as `value`).  Ranks take different tracks (time/track sharding, no collective): weak
scaling, value = N * frames / max-over-ranks step time.

The JSON line also carries
  roofline      : the dominant kernel (raster fill), algorithmic bytes W*H*3 per frame / HIP-event time
  roofline_stft : the fused STFT kernel, algorithmic bytes 4*C*min(hop,1024) + 4*1025 + 8 per frame
  cpu_baseline  : the numpy oracle (a port of the reference's CPU path) timed on one host core over a bounded
                  sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=3600.0, help="track length per GPU per step")
    ap.add_argument("--sample-rate", type=int, default=48000)
    ap.add_argument("--fps", type=int, default=60)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--mode", default="Spectrum - Linear")
    ap.add_argument("--glow", action="store_true", help="closed-path modes: add the glow decoration (7 additive strokes)")
    ap.add_argument("--batch", type=int, default=128, help="frames per raster launch (device slot size)")
    ap.add_argument("--cpu-frames", type=int, default=240, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--pcie-frames", type=int, default=1024, help="frames of the PCIe-inclusive leg (0 = skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="per-launch HBM bytes from separate rocprofv3 --pmc passes (see profiles/README.md)")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (port of export/exporter.py:538-544 + rt_widget.py paint_frame) on one host core."""
    import numpy as np
    from aurora_amd import synth
    from oracle import analysis_ref as A      # checker / baseline only
    from oracle import raster_ref as R

    sr, fps, W, H = args.sample_rate, args.fps, args.width, args.height
    n = int(args.cpu_frames * sr / fps)
    audio = synth.pink(n, 2, seed=1234)
    sunk = 0
    t0 = time.perf_counter()
    ref = A.analyze_track(audio, sr, fps)
    pen, _ = A.amp_to_color_track(ref["energy"])
    F = ref["samples"].shape[0]
    for f in range(F):
        frame = R.render_frame(args.mode, ref["samples"][f], ref["logspec"][f], float(ref["energy"][f]), pen[f], W, H)
        frame = np.ascontiguousarray(frame)            # exporter.py:540
        sunk += len(frame.tobytes())                   # exporter.py:542 -> null sink
    dt = time.perf_counter() - t0
    return {"value": F / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"first {F} frames ({F / fps:.1f} s of audio) of the same workload: numpy oracle analysis + "
                      f"software rasteriser + ascontiguousarray/tobytes into a null sink, {dt:.1f} s on 1 core "
                      f"of {os.cpu_count()} visible"}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share a card for rehearsals on a 1-GPU box
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from aurora_amd import _lib, sharding, synth
    from aurora_amd.engine import GlowConfig, RadialConfig, get_engine, mode_kind
    from aurora_amd.pipeline import FramePipeline, NullSink

    eng = get_engine(dev_index)
    sr, fps, W, H = args.sample_rate, args.fps, args.width, args.height
    n_samples = int(round(args.seconds * sr))
    audio = synth.pink_device(n_samples, 2, 1234 + rank, device)      # input resident in HBM before timing
    track = eng.make_track(audio, sr, fps)
    F = track.total_frames
    # BASELINE.json configs[3]: "particle mode + audio-reactive image feather".  The reference draws the centre disc and
    # its feather only in "Spectrum - Radial" (rt_widget.py:841-849), so this is the union of the work:
    # particle frames with the audio-reactive disc envelope and the six feather rings composited on top.
    union = args.mode == "Particles + Feather"
    kind = mode_kind("Particles" if union else args.mode)
    radial = RadialConfig() if kind == "spectrum_radial" else None
    feather = RadialConfig(feather_enabled=True, feather_audio_enabled=True, feather_audio_amount=60,
                           edge_waviness=40) if union else None
    circular = kind == "waveform_circular"
    planes_mode = radial is not None or circular
    K = {"spectrum_linear": _lib.K_SPECTRUM, "particles": _lib.K_PARTICLES, "waveform_linear": _lib.K_WAVEFORM,
         "spectrum_radial": 0, "waveform_circular": 0}[kind]
    B = max(1, min(args.batch, F))
    slots = 2
    dev_frames = [eng.empty((B, H, W, 3), torch.uint8) for _ in range(slots)]
    if planes_mode:
        dev_spans = [eng.empty((2, B, H, (W + 31) // 32), torch.int32) for _ in range(slots)]
    else:
        dev_spans = [eng.empty((B, K, W), torch.int32) for _ in range(slots)]
    frame_bytes = W * H * 3
    stft_bytes = 4 * track.channels * min(track.hop, 1024) + 4 * 1025 + 8

    def step(ev_fill=None, ev_stft=None):
        if ev_stft is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        an = eng.analyze(track, want_samples=kind.startswith("waveform"))
        if ev_stft is not None:
            e1.record()
            ev_stft.append((e0, e1, F))
        sc = eng.scan(an)
        prep = eng.radial_prepare(an, radial or feather) if (radial or feather) is not None else None
        done, i = 0, 0
        while done < F:
            n = min(B, F - done)
            s = i % slots
            if radial is not None:
                planes = eng.radial_planes(prep, an, sc, W, H, done, n, planes_out=dev_spans[s][:, :n])
            elif circular:
                planes = eng.wave_circular_planes(an, sc, W, H, done, n, planes_out=dev_spans[s][:, :n])
            else:
                spans, _ = eng.spans(kind, an, sc, W, H, done, n, out=dev_spans[s][:n])
                work = eng.summarize(spans, K, W, H)           # column summaries: their own small kernel
            if ev_fill is not None:                            # the events bracket the fill kernel alone
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if planes_mode:
                eng.fill_planes(planes, sc.pen_rgb[done:done + n], W, H, out=dev_frames[s][:n])
            else:
                eng.fill(spans, K, sc.pen_rgb[done:done + n], W, H, out=dev_frames[s][:n], work=work, work_ready=True)
            if ev_fill is not None:
                e1.record()
                ev_fill.append((e0, e1, n))
            if args.glow and planes_mode:
                verts = (eng.radial_path_vertices(prep, sc, W, H, done, n) if radial is not None
                         else eng.wave_circular_vertices(an, sc, W, H, done, n))
                eng.apply_glow(GlowConfig(enabled=True), verts, sc, done, n, planes[0], planes[1], planes[2],
                               dev_frames[s][:n])
            if feather is not None:
                eng.feather_blend(prep, an, sc, W, H, done, n, dev_frames[s][:n], with_disc=True)
            done += n
            i += 1

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    ev_fill, ev_stft = [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(ev_fill, ev_stft)
    barrier()
    dt = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt, device if backend == "nccl" else None)

    def kernel_stats(events, bytes_per_frame):
        ms = sum(a.elapsed_time(b) for a, b, _ in events)
        frames = sum(n for _, _, n in events)
        launches = len(events)
        if launches == 0 or ms <= 0:
            return None
        gbs = frames * bytes_per_frame / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                "avg_launch_ms": round(ms / launches, 4), "frames_per_launch": round(frames / launches, 1),
                "algorithmic_bytes_per_frame": bytes_per_frame}

    roof = kernel_stats(ev_fill, frame_bytes)
    roof_stft = kernel_stats(ev_stft, stft_bytes)
    if roof is not None:
        lin = W % 16 == 0 and os.environ.get("AUR_FILL_MODE", "0") != "1"
        roof["kernel"] = ("raster_fill_planes_kernel" if planes_mode else
                          f"raster_fill_lin_kernel<{K}>" if lin else f"raster_fill16_kernel<{K}>")
    if roof_stft is not None:
        roof_stft["kernel"] = (f"stft_frames_kernel<{track.channels}, log=true, bandmax=false, "
                               f"samples={'true' if kind.startswith('waveform') else 'false'}>")
    try:                                               # per-launch HBM bytes from separate --pmc passes, if collected
        with open(args.traffic_json) as fh:
            tj = json.load(fh)
        key = f"{W}x{H}x{B}"
        if roof is not None and not planes_mode and tj.get("raster_fill", {}).get("config") == key:
            roof["traffic"] = tj["raster_fill"]["hbm_bytes_per_launch"]
        if roof_stft is not None and tj.get("stft", {}).get("frames") == F:
            roof_stft["traffic"] = tj["stft"]["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass

    pcie_fps = pcie_yuv_fps = None
    if args.pcie_frames > 0 and rank == 0:
        nf = min(F, args.pcie_frames)
        for wire in ("rgb24", "yuv420p"):          # the reference's wire format, then the opt-in 1.5 B/px one
            pipe = FramePipeline(eng, track, args.mode, W, H, frame_begin=0, frame_end=nf, batch_frames=min(64, nf),
                                 slots=3, wire_format=wire)
            pipe.analyse()
            sink = NullSink()
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            pipe.run(sink)
            fps_w = nf / (time.perf_counter() - t1)
            assert sink.bytes == nf * pipe.frame_bytes
            if wire == "rgb24":
                pcie_fps = fps_w
            else:
                pcie_yuv_fps = fps_w

    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        cpu = cpu_baseline(args)

    if rank == 0:
        value = world * F * args.steps / dt
        out = {
            "metric": "exported frames/sec (whole node) 4K60 spectrum mode at 1/2/4/8 GPUs; STFT GB/s",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64 FFT / f32 magnitudes / u8 RGB24",
            "data": "synthetic",
            "config": {"workload": f"configs[4] unit: {args.seconds:.0f} s {sr} Hz stereo pink-noise track per GPU, "
                                   f"{args.mode}{' + glow' if args.glow else ''} {W}x{H}@{fps}, {F} frames per GPU per step, frames stay in HBM",
                       "frames_per_gpu": F, "batch_frames": B, "parallelism": f"time/track shards x{world}"},
            "stft_gbs": roof_stft["achieved"] if roof_stft else None,
            "roofline": roof, "roofline_stft": roof_stft, "cpu_baseline": cpu,
            "pcie_inclusive_fps": round(pcie_fps, 1) if pcie_fps else None,
            "pcie_inclusive_fps_yuv420p": round(pcie_yuv_fps, 1) if pcie_yuv_fps else None,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
