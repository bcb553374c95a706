#!/usr/bin/env python3
"""Benchmark of the MI355X export hot path (BASELINE.json metric: exported frames/s, 4K60 spectrum mode; STFT GB/s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`value` is what BASELINE.json names: frames per second that REACH THE HOST SINK.  A step is one whole export of a
synthetic track that is already resident in HBM (default: 240 s of 48 kHz stereo pink noise, "Spectrum - Linear"
3840x2160@60 = 14 400 frames, the track length of configs[1..3] at the headline mode of configs[4]): fused STFT
analysis -> smoothing scan -> column spans -> RGB24 fill in batches -> hipMemcpyAsync into a pinned host ring on a
side stream -> a writer thread hands every slot to the sink (a null sink: FFmpeg is excluded on the GPU and the CPU
side alike).  The ring is allocated before the clock starts.  Ranks export different tracks (time/track sharding, no
collective): weak scaling, value = N * frames / max-over-ranks time.  That rate is bound by the PCIe link of each GPU
(`roofline_pcie`), not by HBM.

Next to it the JSON line carries
  roofline      : the dominant kernel of the step (raster fill), algorithmic bytes W*H*3 per frame / HIP-event time
                  around every fill launch of the timed region
  roofline_pcie : bytes per frame over the link vs its 63 GB/s
  device_leg    : the same path with the frames left in HBM (a 2-slot ring that is overwritten), over the configs[4]
                  unit (1 h track, 216 000 frames per step): `device_fps`, and `roofline_stft` -- the fused STFT kernel
                  over 216 000 frames per launch, algorithmic bytes 4*C*min(hop,1024) + 4*1025 + 8 per frame, against
                  HBM and against the vector-ALU issue roof that actually bounds it
  cpu_baseline  : the numpy oracle (a port of the reference's CPU path) timed on one host core -- how the reference
                  runs -- and on all cores of the box's share, over a bounded sample of the same workload (rank 0, N=1)
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
HBM_WRITE_CEILING_GBS = 6900.0 # tools/write_ceiling.py / tools/fill_lab on this pool: what torch's fill_ / zero_ and a bare store kernel reach
PCIE_PEAK_GBS = 63.0           # same guide: host link PCIe Gen5 x16, 63 GB/s
PCIE_COPY_GBS = 57.1           # profiles/r02_d2h_probe.txt: a bare hipMemcpyAsync of one ring slot into pinned memory on this pool
F64_ISSUE_NS = 2.18            # profiles/r01_f64_issue_probe.txt: ns per VALU wave-instruction per SIMD under this load


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=240.0, help="track length per GPU per step (the exported track)")
    ap.add_argument("--sample-rate", type=int, default=48000)
    ap.add_argument("--fps", type=int, default=60)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--mode", default="Spectrum - Linear")
    ap.add_argument("--signal", default="pink", choices=("pink", "sweep"),
                    help="synthetic track: pink noise (BASELINE.json configs[1..4]) or the 20 Hz - 20 kHz sine sweep (configs[0])")
    ap.add_argument("--glow", action="store_true", help="closed-path modes: add the glow decoration (7 additive strokes)")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per raster launch / ring slot (0 = as many as fit the 1.6 GB of 64 4K frames: small "
                         "frames in small batches leave the device leg bound by the host's launch rate)")
    ap.add_argument("--slots", type=int, default=3, help="ring slots in flight")
    ap.add_argument("--wire", default="rgb24", choices=("rgb24", "yuv420p", "nv12"), help="wire format of the headline leg")
    ap.add_argument("--device-seconds", type=float, default=3600.0, help="track of the device-side leg (configs[4] unit)")
    ap.add_argument("--device-steps", type=int, default=4, help="timed steps of the device-side leg (0 = skip)")
    ap.add_argument("--device-warmup", type=int, default=1, help="untimed steps of the device-side leg")
    ap.add_argument("--cpu-frames", type=int, default=240, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the all-cores CPU figure (0 = box share)")
    ap.add_argument("--yuv-leg", type=int, default=1, help="also time one export step in the opt-in yuv420p wire format")
    ap.add_argument("--events", type=int, default=1,
                    help="0: record no HIP events around the device leg's kernels (counter passes: tools/pmc_passes.sh)")
    ap.add_argument("--pipe-frames", type=int, default=1920,
                    help="frames of the pipe leg: the same export through the product's pipe sink into a child process "
                         "that reads stdin as FFmpeg does (0 = skip)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="per-launch HBM bytes from separate rocprofv3 --pmc passes (see profiles/README.md)")
    ap.add_argument("--isa-json", default=os.path.join(ROOT, "profiles", "stft_isa.json"),
                    help="VALU instructions per frame of the STFT loop (tools/stft_isa.py)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------- CPU baseline

def _cpu_export(job):
    """The oracle (port of export/exporter.py:538-544 + rt_widget.py paint_frame) over `frames` frames -> seconds."""
    import numpy as np
    from aurora_amd import synth
    from oracle import analysis_ref as A      # checker / baseline only
    from oracle import raster_ref as R
    mode, sr, fps, W, H, frames, seed, signal = job[:8]
    piped = len(job) > 8 and job[8]
    n = int(frames * sr / fps)
    audio = synth.sweep(n, sr, 2) if signal == "sweep" else synth.pink(n, 2, seed=seed)
    sunk = 0
    child = None
    if piped:                                          # exporter.py:532-535: the reference's Popen, pipe left at 64 KiB
        import subprocess
        child = subprocess.Popen(_drain_cmd(), stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=10 ** 7)
    t0 = time.perf_counter()
    ref = A.analyze_track(audio, sr, fps)
    pen, _ = A.amp_to_color_track(ref["energy"])
    F = ref["samples"].shape[0]
    for f in range(F):
        frame = R.render_frame(mode, ref["samples"][f], ref["logspec"][f], float(ref["energy"][f]), pen[f], W, H)
        frame = np.ascontiguousarray(frame)            # exporter.py:540
        if child is not None:
            child.stdin.write(frame.tobytes())         # exporter.py:542
        else:
            sunk += len(frame.tobytes())               # exporter.py:542 -> null sink
    if child is not None:
        child.stdin.close()
        rep = json.loads(child.stdout.read())
        child.wait()
        assert rep["bytes"] == F * W * H * 3
    return F, time.perf_counter() - t0


def _drain_cmd():
    return [sys.executable, os.path.join(ROOT, "tools", "pipe_drain.py")]


def cpu_baseline(args):
    import multiprocessing as mp
    sr, fps, W, H = args.sample_rate, args.fps, args.width, args.height
    F, dt = _cpu_export((args.mode, sr, fps, W, H, args.cpu_frames, 1234, args.signal))
    out = {"value": round(F / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"first {F} frames ({F / fps:.1f} s of audio) of the same workload: numpy oracle analysis + software "
                     f"rasteriser + ascontiguousarray/tobytes into a null sink, {dt:.1f} s on 1 core (the reference "
                     f"exports in one process, export/worker.py)"}
    if args.pipe_frames > 0:                         # the boundary the reference ends at: proc.stdin.write into a reading child
        per = max(30, args.cpu_frames // 2)
        Fp, dtp = _cpu_export((args.mode, sr, fps, W, H, per, 1234, args.signal, True))
        out["piped"] = {"value": round(Fp / dtp, 2), "unit": "frames/s", "cores": 1,
                        "sample": f"first {Fp} frames, frame.tobytes() written to the stdin of a child that reads and "
                                  f"discards (the reference's Popen: bufsize 10**7, 64 KiB pipe), {dtp:.1f} s"}
    # courtesy whole-host figure: independent processes over disjoint time ranges of the track
    procs = args.cpu_procs or min(16, os.cpu_count() or 1)      # a 1-GPU box's CPU share is 16 (nproc shows the machine)
    if procs > 1:
        per = max(30, args.cpu_frames // 2)
        ctx = mp.get_context("spawn")                            # this process has initialised the GPU: never fork it
        t0 = time.perf_counter()
        with ctx.Pool(procs) as pool:
            res = pool.map(_cpu_export, [(args.mode, sr, fps, W, H, per, 1234 + 7 * k, args.signal) for k in range(procs)])
        wall = time.perf_counter() - t0
        busy = max(r[1] for r in res)
        out["all_cores"] = {"value": round(sum(r[0] for r in res) / busy, 2), "unit": "frames/s", "cores": procs,
                            "sample": f"{procs} processes x {per} frames of disjoint ranges, slowest process {busy:.1f} s "
                                      f"({wall:.1f} s with interpreter start-up)"}
    return out


# ---------------------------------------------------------------------------------------------------- helpers

def kernel_stats(events, bytes_per_frame):
    ms = sum(a.elapsed_time(b) for a, b, _ in events)
    frames = sum(n for _, _, n in events)
    launches = len(events)
    if launches == 0 or ms <= 0:
        return None
    gbs = frames * bytes_per_frame / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "write_ceiling_gbs": HBM_WRITE_CEILING_GBS, "frac_of_write_ceiling": round(gbs / HBM_WRITE_CEILING_GBS, 4),
            "avg_launch_ms": round(ms / launches, 4), "frames_per_launch": round(frames / launches, 1),
            "algorithmic_bytes_per_frame": bytes_per_frame}


def main():
    args = parse_args()
    # multi-process GPU work on this pool needs dmabuf IPC (already exported there; kept if the launcher dropped it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process has imported nothing that touches a GPU, so it
        # starts N fresh ranks of itself (one per GPU), forwards rank 0's JSON line and exits with the worst status
        from aurora_amd.sharding import launch_ranks
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share a card for rehearsals on a 1-GPU box
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    # before the HIP runtime starts (device_count does not start it): this rank and every thread it will ever have on the
    # CPUs next to its GPU, from the KFD topology in sysfs; hosts that hide it are handled after the runtime is up
    from aurora_amd import sharding as _sh
    early = _sh.bind_near_device(None, dev_index)
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
    # threads and the pinned frame ring of this rank on the socket its GPU hangs off (a no-op where the host hides it)
    placement = early if early["bus_id"] else sharding.bind_near_device(eng.lib, dev_index)
    sr, fps, W, H = args.sample_rate, args.fps, args.width, args.height
    frame_bytes = W * H * 3
    if args.batch <= 0:
        args.batch = int(max(8, min(1024, (64 * 3840 * 2160 * 3) // frame_bytes)))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    t_start = time.perf_counter()

    def note(msg):
        # progress on stderr (stdout carries the one JSON line): a long run must not look hung to whoever watches it
        if rank == 0:
            print(f"[bench {time.perf_counter() - t_start:6.1f} s] {msg}", file=sys.stderr, flush=True)

    # BASELINE.json configs[3]: "particle mode + audio-reactive image feather".  The reference draws the centre disc and
    # its feather only in "Spectrum - Radial" (rt_widget.py:841-849), so this is the union of the work:
    # particle frames with the audio-reactive disc envelope and the six feather rings composited on top (device leg only).
    union = args.mode == "Particles + Feather"
    kind = mode_kind("Particles" if union else args.mode)
    planes_mode = kind in ("spectrum_radial", "waveform_circular")
    K = {"spectrum_linear": _lib.K_SPECTRUM, "particles": _lib.K_PARTICLES, "waveform_linear": _lib.K_WAVEFORM,
         "spectrum_radial": 0, "waveform_circular": 0}[kind]

    # ------------------------------------------------------------------ headline: export into the host sink
    n_samples = int(round(args.seconds * sr))
    def make_audio(n):
        if args.signal == "sweep":
            return synth.sweep_device(n, sr, 2, device)
        return synth.pink_device(n, 2, 1234 + rank, device)

    audio = make_audio(n_samples)                                     # input resident in HBM before timing
    track = eng.make_track(audio, sr, fps)
    F = track.total_frames
    stft_bytes = 4 * track.channels * min(track.hop, 1024) + 4 * 1025 + 8
    export_mode = "Particles" if union else args.mode
    B = max(1, min(args.batch, F))

    def make_pipe(wire):
        return FramePipeline(eng, track, export_mode, W, H, batch_frames=B, slots=args.slots, wire_format=wire,
                             glow=GlowConfig(enabled=True) if args.glow else None).prepare()

    def export_step(pipe):
        pipe.reset_analysis()                      # a step is the whole job: analysis + scan + raster + readback
        sink = NullSink()
        n = pipe.run(sink)
        assert n == F and sink.bytes == F * pipe.frame_bytes
        return n

    if args.steps <= 0:                            # counter passes profile the device leg alone: no export leg
        args.warmup, args.yuv_leg = 0, 0
    roof, wire_bytes, dt = None, frame_bytes, 1.0
    if args.steps > 0:
        pipe = make_pipe(args.wire)                    # device slots + pinned ring allocated here, before the clock
        for k in range(args.warmup):
            export_step(pipe)
            note(f"export warm-up step {k + 1}/{args.warmup}")
        ev_fill, ev_stft_small = [], []
        if not planes_mode:
            pipe.fill_events = ev_fill
        pipe.stft_events = ev_stft_small
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            export_step(pipe)
            note(f"export step {k + 1}/{args.steps}")
        barrier()
        dt = time.perf_counter() - t0
        dt = sharding.max_over_ranks(dt, device if backend == "nccl" else None)
        roof = kernel_stats(ev_fill, frame_bytes)
        wire_bytes = pipe.frame_bytes
        pipe.fill_events = pipe.stft_events = None
        pipe.close()
        del pipe
    value = world * F * args.steps / dt if args.steps > 0 else None
    link_gbs = F * args.steps / dt * wire_bytes / 1e9                 # per GPU (the slowest rank's)

    yuv_fps = None
    if args.yuv_leg and rank == 0 and args.wire == "rgb24" and W % 4 == 0 and H % 2 == 0:
        p2 = make_pipe("yuv420p")
        export_step(p2)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        export_step(p2)
        yuv_fps = F / (time.perf_counter() - t1)
        yuv_bytes = p2.frame_bytes
        p2.close()
        del p2
        note("yuv420p leg done")
    # ------------------------------------------------------------------ pipe leg: the boundary the reference ends at
    # export/exporter.py:532-542 ends at proc.stdin.write: the same export through the product's own pipe sink
    # (aurora_amd.export.exporter._PipeSink behind _grow_pipe) into a fresh child that reads stdin and discards
    pipe_leg = None
    if args.pipe_frames > 0 and rank == 0 and args.steps > 0:
        import subprocess
        from aurora_amd.export.exporter import _PipeSink, _grow_pipe
        PF = min(F, args.pipe_frames)

        def piped(wire, splice):
            p = FramePipeline(eng, track, export_mode, W, H, frame_end=PF, batch_frames=min(B, PF), slots=args.slots,
                              wire_format=wire, glow=GlowConfig(enabled=True) if args.glow else None).prepare()
            child = subprocess.Popen(_drain_cmd(), stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=10 ** 7)
            cap = _grow_pipe(child.stdin)
            sink = _PipeSink(child.stdin, splice=splice)
            torch.cuda.synchronize(device)
            t = time.perf_counter()
            n = p.run(sink)
            child.stdin.close()
            rep = json.loads(child.stdout.read())
            child.wait()
            t = time.perf_counter() - t
            p.close()
            sink.close()
            assert n == PF and rep["bytes"] == PF * p.frame_bytes
            how = ("write" if sink.spliced_bytes == 0 else
                   "staged vmsplice (ring memory refused: errno %s)" % sink.refused_errno if sink._staged else "vmsplice")
            return {"fps": round(PF / t, 1), "gbs": round(PF * p.frame_bytes / t / 1e9, 2), "bytes_per_frame": p.frame_bytes,
                    "pipe_bytes": cap, "path": how, "spliced_frac": round(sink.spliced_bytes / max(1, rep["bytes"]), 4)}

        pipe_leg = {"frames": PF, "reader": "tools/pipe_drain.py: read() of 4 MiB into its own buffer, discards"}
        pipe_leg["rgb24"] = piped("rgb24", None)
        note("pipe leg: rgb24 done")
        pipe_leg["rgb24_plain_write"] = piped("rgb24", False)
        note("pipe leg: rgb24 with plain write done")
        if W % 4 == 0 and H % 2 == 0:
            pipe_leg["yuv420p"] = piped("yuv420p", None)
            note("pipe leg: yuv420p done")
    del track, audio
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ device-side leg (frames stay in HBM)
    device_leg = None
    roof_stft = None
    if args.device_steps > 0:
        dn = int(round(args.device_seconds * sr))
        daudio = make_audio(dn)
        dtrack = eng.make_track(daudio, sr, fps)
        DF = dtrack.total_frames
        radial = RadialConfig() if kind == "spectrum_radial" else None
        feather = RadialConfig(feather_enabled=True, feather_audio_enabled=True, feather_audio_amount=60,
                               edge_waviness=40) if union else None
        circular = kind == "waveform_circular"
        DB = max(1, min(args.batch, DF))
        slots = 2
        dev_frames = [eng.empty((DB, H, W, 3), torch.uint8) for _ in range(slots)]
        if planes_mode:
            dev_spans = [eng.empty((2, DB, H, (W + 31) // 32), torch.int32) for _ in range(slots)]
        else:
            dev_spans = [eng.empty((DB, K, W), torch.int32) for _ in range(slots)]

        held = {"an": None}           # the analysis buffers of the previous step are overwritten: no allocation between the events

        def step(ev_f=None, ev_s=None):
            if ev_s is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            an = held["an"] = eng.analyze(dtrack, want_samples=kind.startswith("waveform"), reuse=held["an"])
            if ev_s is not None:
                e1.record()
                ev_s.append((e0, e1, DF))
            sc = eng.scan(an)
            prep = eng.radial_prepare(an, radial or feather) if (radial or feather) is not None else None
            done, i = 0, 0
            while done < DF:
                n = min(DB, DF - done)
                s = i % slots
                if planes_mode:
                    # the whole frame as an export renders it (stroke plane, two-pass fill, decorations, antialiased
                    # disc); the events the engine records bracket the plane fill (both passes) alone
                    eng.fill_events = ev_f
                    glow = GlowConfig(enabled=True) if args.glow else None
                    if radial is not None:
                        eng.render_radial(prep, an, sc, W, H, done, n, out=dev_frames[s][:n],
                                          planes_out=dev_spans[s][:, :n], glow=glow)
                    else:
                        eng.render_wave_circular(an, sc, W, H, done, n, out=dev_frames[s][:n],
                                                 planes_out=dev_spans[s][:, :n], glow=glow)
                    eng.fill_events = None
                else:
                    spans, _ = eng.spans(kind, an, sc, W, H, done, n, out=dev_spans[s][:n])
                    work = eng.summarize(spans, K, W, H)           # column summaries: their own small kernel
                    if ev_f is not None:                           # the events bracket the fill kernel alone
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    eng.fill(spans, K, sc.pen_rgb[done:done + n], W, H, out=dev_frames[s][:n], work=work, work_ready=True)
                    if ev_f is not None:
                        e1.record()
                        ev_f.append((e0, e1, n))
                if feather is not None:
                    eng.feather_blend(prep, an, sc, W, H, done, n, dev_frames[s][:n], with_disc=True)
                done += n
                i += 1

        for _ in range(args.device_warmup):
            step()
        note("device leg warm-up done")
        ev_f, ev_s = [], []
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.device_steps):
            step(ev_f, ev_s) if args.events else step()
        barrier()
        ddt = sharding.max_over_ranks(time.perf_counter() - t1, device if backend == "nccl" else None)
        note("device leg done")
        roof_dev = kernel_stats(ev_f, frame_bytes)
        roof_stft = kernel_stats(ev_s, stft_bytes)
        if roof_stft is not None:                      # a read-and-compute kernel: the store ceiling says nothing about it
            roof_stft.pop("write_ceiling_gbs", None)
            roof_stft.pop("frac_of_write_ceiling", None)
        # `roofline` is the fill kernel at steady clocks (back-to-back launches of the device leg).  Inside the export
        # steps the same launches are timed too (`fill_in_export_steps`): there the GPU idles ~98 % of the time waiting
        # for the PCIe link and its clocks ramp down between the 0.24 ms bursts, so the kernel runs slower without
        # costing the export anything.
        in_export = roof
        roof = roof_dev
        if roof is not None:
            roof["measured_in"] = "device_leg timed region"
            roof["fill_in_export_steps"] = ({k: in_export[k] for k in ("achieved", "frac", "avg_launch_ms",
                                                                        "frames_per_launch")} if in_export else None)
        device_leg = {"device_fps": round(world * DF * args.device_steps / ddt, 1),
                      "ms_per_step": round(ddt / args.device_steps * 1e3, 3), "frames_per_gpu_per_step": DF,
                      "workload": f"configs[4] unit: {args.device_seconds:.0f} s track per GPU, frames stay in HBM "
                                  f"(2 slots of {DB} frames, overwritten)",
                      }
        if roof_stft is not None:
            roof_stft["kernel"] = (f"stft_frames_kernel<{dtrack.channels}, log=true, bandmax=false, "
                                   f"samples={'true' if kind.startswith('waveform') else 'false'}>")
            # what bounds this kernel is the vector ALU, not HBM (traffic = 1.07 x algorithmic, VALU busy 74 %): every
            # VALU wave-instruction, float64 or not, occupies its SIMD for 4 cycles
            try:
                with open(args.isa_json) as fh:
                    valu = int(json.load(fh)["valu_per_frame"])
                num_simd = 4 * torch.cuda.get_device_properties(device).multi_processor_count
                issue_ms = DF / num_simd * valu * F64_ISSUE_NS * 1e-6
                roof_stft.update({"bound": "valu", "valu_insts_per_frame": valu, "valu_issue_ns_per_inst": F64_ISSUE_NS,
                                  "issue_roof_ms": round(issue_ms, 4),
                                  "frac_of_issue_roof": round(issue_ms / roof_stft["avg_launch_ms"], 4),
                                  "frac_of_hbm_peak": roof_stft["frac"]})
            except (OSError, ValueError, KeyError):
                pass
        del dev_frames, dev_spans, dtrack, daudio, held

    if roof is not None:
        lin = W % 16 == 0 and os.environ.get("AUR_FILL_MODE", "0") != "1"
        if planes_mode:
            # the label follows the path the device leg took (the events bracket that call): without decorations the
            # engine paints the stroke straight into the frames, with them it goes through the stroke plane
            direct = eng._direct_stroke_ok(W, H, DB if args.device_steps > 0 else B, eng.empty((1, 16, 64, 3), torch.uint8),
                                           bool(args.glow))
            roof["kernel"] = ("raster_fill_rows_kernel + paint_stroke_closed_kernel (aur_raster_closed_path)" if direct else
                              "raster_fill_rows_kernel + raster_patch_planes_kernel (aur_raster_planes, both passes)"
                              if W % 32 == 0 and lin else "raster_fill_planes_kernel")
        else:
            roof["kernel"] = f"raster_fill_lin_kernel<{K}>" if lin else f"raster_fill16_kernel<{K}>"
    try:                                               # per-launch HBM bytes from separate --pmc passes, if collected
        with open(args.traffic_json) as fh:
            tj = json.load(fh)
        rf = tj.get("raster_fill", {})
        if roof is not None and not planes_mode and rf.get("config") == f"{W}x{H}" and rf.get("hbm_bytes_per_launch"):
            # per-launch bytes from tools/pmc_passes.sh, collected on 128-frame launches (one of the dispatch sequences
            # rocprofv3's counter mode survives here, profiles/r03_pmc_fault_probe.txt); the fill's traffic is linear in
            # the frames of a launch, so it is scaled to this run's launch size
            roof["traffic"] = int(round(rf["hbm_bytes_per_launch"] * roof["frames_per_launch"] / rf["frames_per_launch"]))
        if roof_stft is not None and tj.get("stft", {}).get("frames") == roof_stft["frames_per_launch"]:
            roof_stft["traffic"] = tj["stft"]["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass

    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        cpu = cpu_baseline(args)
        note("CPU baseline done")

    if rank == 0:
        out = {
            "metric": "exported frames/sec (whole node) 4K60 spectrum mode at 1/2/4/8 GPUs; STFT GB/s",
            "value": round(value, 1) if value is not None else None, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64 FFT / f32 magnitudes / u8 RGB24",
            "data": "synthetic",
            "config": {"workload": f"{args.seconds:.0f} s {sr} Hz stereo {'sine-sweep' if args.signal == 'sweep' else 'pink-noise'} track per GPU "
                                   f"(resident in HBM), "
                                   f"{args.mode}{' + glow' if args.glow else ''} {W}x{H}@{fps}, {F} frames per GPU per "
                                   f"step, every frame read back through a pinned ring into a null sink ({args.wire})",
                       "frames_per_gpu": F, "batch_frames": B, "ring_slots": args.slots, "wire_format": args.wire,
                       "parallelism": f"time/track shards x{world}", "host_placement_rank0": placement},
            "stft_gbs": roof_stft["achieved"] if roof_stft else None,
            "roofline": roof,
            "roofline_pcie": {"bound": "pcie", "achieved": round(link_gbs, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                              "frac": round(link_gbs / PCIE_PEAK_GBS, 4), "bytes_per_frame": wire_bytes,
                              "bare_copy_gbs": PCIE_COPY_GBS,
                              "note": "per GPU: every exported frame crosses the GPU's PCIe Gen5 x16 link once"},
            "roofline_stft": roof_stft,
            "device_fps": device_leg["device_fps"] if device_leg else None,
            "device_leg": device_leg,
            "yuv420p_leg": ({"fps": round(yuv_fps, 1), "link_gbs": round(yuv_fps * yuv_bytes / 1e9, 2),
                             "bytes_per_frame": yuv_bytes} if yuv_fps else None),
            "pipe_leg": pipe_leg,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
