"""Frame pipeline: analysis -> scan -> rasterise in batches on the GPU, pinned async readback on a side
stream, ordered hand-off to a sink (FFmpeg stdin, a file, or a null sink) from a writer thread.

Replaces the per-frame loop of export/exporter.py:538-549 and its three host copies
(``arr.copy()`` :95, ``tobytes()`` :542, pipe write): frames leave the GPU once, into page-locked memory,
and are written to the sink straight from there.

Time sharding (one process per GPU): a pipeline renders frames ``[frame_begin, frame_end)`` of a track.  The
analysis halo (previous block / previous magnitudes) is rebuilt inside the STFT kernel; the carried smoothing
state (colour EMA, flux EMA, phase) comes from an analysis-only pre-pass over ``[0, frame_begin)``
(``HipExportEngine.carried_state``), so a shard's bytes are identical to the same frames of an unsharded run.
"""
from __future__ import annotations

import ctypes as C
import queue
import threading

import numpy as np

from . import _lib
from .engine import FillConfig, GlowConfig, ShadowConfig, HipExportEngine, PLANE_KINDS, RadialConfig, Track, mode_kind, SUPPORTED_KINDS


class NullSink:
    """Counts bytes; stands in for FFmpeg in benchmarks (FFmpeg is excluded on both CPU and GPU sides)."""

    def __init__(self):
        self.bytes = 0
        self.frames = 0

    def write(self, buf):
        self.bytes += len(buf)

    def close(self):
        pass


class FileSink:
    def __init__(self, path):
        self._f = open(path, "wb")

    def write(self, buf):
        self._f.write(buf)

    def close(self):
        self._f.close()


class _PinnedSlot:
    def __init__(self, lib, nbytes):
        self.lib = lib
        p = C.c_void_p()
        _lib.check(lib.aur_host_alloc(C.byref(p), nbytes), "aur_host_alloc")
        self.ptr = p
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def free(self):
        if self.ptr:
            self.lib.aur_host_free(self.ptr)
            self.ptr = None


def default_batch_frames(W: int, H: int, budget_bytes: int = 1 << 30) -> int:
    """Frames per batch so one device slot is about ``budget_bytes`` (at least 8, at most 1024 frames)."""
    return int(max(8, min(1024, budget_bytes // (W * H * 3))))


class FramePipeline:
    def __init__(self, engine: HipExportEngine, track: Track, mode: str, width: int, height: int,
                 sensitivity: float = 1.0, frame_begin: int = 0, frame_end: int | None = None,
                 batch_frames: int | None = None, slots: int = 2, scan_params: _lib.ScanParams | None = None,
                 radial: "RadialConfig | None" = None, rotation_deg: float = 0.0, wire_format: str = "rgb24",
                 fill: "FillConfig | None" = None, shadow: "ShadowConfig | None" = None,
                 glow: "GlowConfig | None" = None, background=None):
        self.eng = engine
        self.track = track
        self.mode = mode
        self.kind = mode_kind(mode)
        if self.kind not in SUPPORTED_KINDS:
            raise NotImplementedError(f"mode {mode!r} ({self.kind}) is not implemented on the HIP path yet")
        self.W, self.H = int(width), int(height)
        self.frame_begin = int(frame_begin)
        self.frame_end = track.total_frames if frame_end is None else int(frame_end)
        if not (0 <= self.frame_begin <= self.frame_end <= track.total_frames):
            raise ValueError("bad frame range")
        self.batch = int(batch_frames or default_batch_frames(self.W, self.H))
        self.slots = int(max(1, slots))
        self.params = scan_params or _lib.ScanParams.defaults(sensitivity)
        if wire_format not in ("rgb24", "yuv420p", "nv12"):
            raise ValueError(f"wire_format {wire_format!r}: expected 'rgb24' (the reference's, exporter.py:429-434), "
                             "'yuv420p' or 'nv12'")
        if wire_format != "rgb24" and (self.W % 4 or self.H % 2):
            raise ValueError(f"{wire_format} wire format needs a width that is a multiple of 4 and an even height")
        self.wire_format = wire_format
        self.rgb_bytes = self.W * self.H * 3
        # bytes per frame on the wire (readback, sink)
        self.frame_bytes = self.rgb_bytes if wire_format == "rgb24" else self.rgb_bytes // 2
        self.radial = (radial or RadialConfig.from_view_state(None, sensitivity)) if self.kind == "spectrum_radial" else None
        self.rotation_deg = float(rotation_deg)          # radial_rotation_deg, used by "Waveform - Circular" too
        self.planes = self.kind in PLANE_KINDS
        self.fill = fill if (fill is not None and fill.enabled and self.planes) else None    # `_apply_fill`: closed paths only
        # the reference draws no shadow for the bar spectrum or the particles; the linear waveform has its own
        # RoundCap / RoundJoin variant (rt_widget.py:915-930)
        self.shadow = shadow if (shadow is not None and shadow.active and
                                 (self.planes or self.kind == "waveform_linear")) else None
        # `_draw_glow_path` is only called for the closed paths (rt_widget.py:901, :980)
        self.glow = glow if (glow is not None and glow.active and self.planes) else None
        # background image (`_draw_background`, rt_widget.py:505-545): composed once, every frame starts from it
        self.background = background if (background is not None and background.active) else None
        self._base = None
        self._base_done = False
        self._analysed = None
        self._radial_prep = None
        self._ring = None
        self.fill_events = None      # benchmarks set a list: (start, end, n_frames) HIP events around every fill launch
        self.stft_events = None      # the same around the fused STFT launch of analyse()

    # ---- phase 1: analysis + scan of the whole shard (about 10 KB per frame) -------------------------
    def analyse(self):
        if self._analysed is None:
            eng = self.eng
            state = eng.carried_state(self.track, self.frame_begin, self.params) if self.frame_begin > 0 else None
            n = self.frame_end - self.frame_begin
            if self.stft_events is not None:
                e0, e1 = eng.torch.cuda.Event(enable_timing=True), eng.torch.cuda.Event(enable_timing=True)
                e0.record()
            an = eng.analyze(self.track, self.frame_begin, n, want_samples=self.kind.startswith("waveform"))
            if self.stft_events is not None:
                e1.record()
                self.stft_events.append((e0, e1, n))
            sc = eng.scan(an, self.params, state)
            if self.radial is not None:
                rstate = eng.carried_radial_state(self.track, self.frame_begin, self.radial)
                self._radial_prep = eng.radial_prepare(an, self.radial, rstate)
            self._analysed = (an, sc)
        return self._analysed

    # ---- device-only rendering (benchmark "value": frames stay in HBM) ----------------------------------
    def render_device(self, out=None, spans_out=None, first: int = 0, count: int | None = None):
        an, sc = self.analyse()
        return self._render(an, sc, first, count, out, spans_out)

    def _render(self, an, sc, first, count, out=None, scratch=None):
        if self.background is not None and not self._base_done:
            self._base = self.eng.compose_background(self.background, self.W, self.H)
            self._base_done = True
        # the base frame is a property of the context: every render states whose it is (pipelines share engines)
        self.eng.set_background(self._base, self.W, self.H)
        try:
            return self._render_batch(an, sc, first, count, out, scratch)
        finally:
            if self._base is not None:
                self.eng.set_background(None)

    def render_one(self, an, sc, radial_prep=None):
        """uint8 [1, H, W, 3] of ONE analysed frame whose analysis, scan and ring state the caller stepped itself
        (`HipOffscreenRenderer.render_time` off the frame grid: the reference's call-by-call semantics)."""
        saved = self._radial_prep
        self._radial_prep = radial_prep
        try:
            return self._render(an, sc, 0, 1)
        finally:
            self._radial_prep = saved

    def _render_batch(self, an, sc, first, count, out=None, scratch=None):
        if self.radial is not None:
            return self.eng.render_radial(self._radial_prep, an, sc, self.W, self.H, first, count, out=out,
                                          planes_out=scratch, fill=self.fill, shadow=self.shadow, glow=self.glow)
        if self.kind == "waveform_linear" and self.shadow is not None:
            n = an.n_frames - first if count is None else count
            return self.eng.render_wave_linear_shadow(an, sc, self.W, self.H, first, n, self.shadow, out=out)
        if self.kind == "waveform_circular":
            return self.eng.render_wave_circular(an, sc, self.W, self.H, first, count, self.rotation_deg, out=out,
                                                 planes_out=scratch, fill=self.fill, shadow=self.shadow, glow=self.glow)
        ev = self.fill_events
        if ev is None:
            return self.eng.render(self.mode, an, sc, self.W, self.H, first, count, out=out, spans_out=scratch)
        # benchmark instrumentation: the same three launches as eng.render, with HIP events around the fill kernel alone
        # (recorded on the stream the kernels are launched on)
        torch = self.eng.torch
        n = an.n_frames - first if count is None else count
        spans, K = self.eng.spans(self.kind, an, sc, self.W, self.H, first, n, out=scratch)
        work = self.eng.summarize(spans, K, self.W, self.H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self.eng.fill(spans, K, sc.pen_rgb[first:first + n], self.W, self.H, out=out, work=work, work_ready=True)
        e1.record()
        ev.append((e0, e1, n))
        return out

    # ---- full pipeline: GPU -> pinned host ring -> sink --------------------------------------------------
    def prepare(self):
        """Allocates the device slots, the pinned host ring and the copy stream once (idempotent).  `run` does it on its
        own when nobody did; a caller that exports repeatedly (benchmarks, many tracks of one size) calls it once and
        `close()` at the end, so that no allocation sits inside a run."""
        if self._ring is not None:
            return self
        eng, torch = self.eng, self.eng.torch
        n_total = max(1, self.frame_end - self.frame_begin)
        B, S = min(self.batch, n_total), self.slots
        dev_frames = [eng.empty((B, self.H, self.W, 3), torch.uint8) for _ in range(S)]
        yuv = self.wire_format != "rgb24"
        dev_wire = [eng.empty((B, self.frame_bytes), torch.uint8) for _ in range(S)] if yuv else dev_frames
        if self.planes:
            dev_spans = [eng.empty((2, B, self.H, (self.W + 31) // 32), torch.int32) for _ in range(S)]
        else:
            K = {"spectrum_linear": _lib.K_SPECTRUM, "particles": _lib.K_PARTICLES,
                 "waveform_linear": _lib.K_WAVEFORM}[self.kind]
            dev_spans = [eng.empty((B, K, self.W), torch.int32) for _ in range(S)]
        host = [_PinnedSlot(eng.lib, B * self.frame_bytes) for _ in range(S)]
        self._ring = dict(B=B, S=S, dev_frames=dev_frames, dev_wire=dev_wire, dev_spans=dev_spans, host=host,
                          copy_stream=torch.cuda.Stream(device=eng.device))
        return self

    def _fused_yuv_ok(self):
        """yuv420p / nv12 straight from the spans: span modes without a background image or a shadow layer."""
        return (not self.planes and self.background is None and self.shadow is None
                and self.W % 4 == 0 and self.H % 2 == 0)

    def close(self):
        """Frees the pinned ring (device tensors go back to torch's allocator with the object)."""
        if self._ring is not None:
            self.eng.torch.cuda.synchronize(self.eng.device)
            for h in self._ring["host"]:
                h.free()
            self._ring = None

    def reset_analysis(self):
        """Forget the analysed shard: the next run / render analyses again (benchmark steps time the whole job)."""
        self._analysed = None
        self._radial_prep = None

    def run(self, sink, progress=None):
        """Renders every frame of the shard in order into ``sink.write``.  Returns the number of frames."""
        eng, torch = self.eng, self.eng.torch
        an, sc = self.analyse()
        n_total = self.frame_end - self.frame_begin
        if n_total == 0:
            return 0
        own_ring = self._ring is None
        self.prepare()
        ring = self._ring
        B, S = ring["B"], ring["S"]
        dev_frames, dev_wire, dev_spans, host = ring["dev_frames"], ring["dev_wire"], ring["dev_spans"], ring["host"]
        copy_stream = ring["copy_stream"]
        yuv = self.wire_format != "rgb24"
        nv12 = self.wire_format == "nv12"
        copied = [None] * S                    # event: slot's D2H copy finished
        free_slots = threading.Semaphore(S)    # host slot consumed by the writer
        q: "queue.Queue" = queue.Queue()
        err = []

        def writer():
            try:
                while True:
                    item = q.get()
                    if item is None:
                        return
                    slot, nframes, ev, first = item
                    ev.synchronize()
                    if not err:
                        try:
                            sink.write(memoryview(host[slot].array)[:nframes * self.frame_bytes])
                        except BrokenPipeError as e:          # exporter.py:543-544: stop writing
                            err.append(e)
                    free_slots.release()
                    if progress is not None and not err:
                        progress(first + nframes, n_total)
            except BaseException as e:   # keep the producer from blocking forever
                err.append(e)
                free_slots.release()

        th = threading.Thread(target=writer, name="aurora-sink-writer", daemon=True)
        th.start()
        done = 0
        try:
            i = 0
            while done < n_total and not err:
                slot = i % S
                n = min(B, n_total - done)
                free_slots.acquire()                       # host slot drained by the writer
                if err:
                    free_slots.release()
                    break
                cur = torch.cuda.current_stream(eng.device)
                if copied[slot] is not None:
                    cur.wait_event(copied[slot])           # device slot no longer being read by the copy engine
                fused = yuv and self._fused_yuv_ok()
                if fused:
                    # span modes: the planar frame straight from the spans, no rgb24 frame in between
                    spans, K = eng.spans(self.kind, an, sc, self.W, self.H, done, n, out=dev_spans[slot][:n])
                    entry = eng.lib.aur_raster_spans_nv12 if nv12 else eng.lib.aur_raster_spans_yuv420p
                    _lib.check(entry(
                        eng.ctx.handle, C.c_void_p(spans.data_ptr()), K, C.c_void_p(sc.pen_rgb[done:done + n].data_ptr()),
                        n, self.W, self.H, C.c_void_p(dev_wire[slot].data_ptr()), eng.stream()),
                        "aur_raster_spans_" + self.wire_format)
                else:
                    self._render(an, sc, done, n, out=dev_frames[slot][:n],
                                 scratch=dev_spans[slot][:, :n] if self.planes else dev_spans[slot][:n])
                if yuv and not fused:
                    entry = eng.lib.aur_rgb24_to_nv12 if nv12 else eng.lib.aur_rgb24_to_yuv420p
                    _lib.check(entry(
                        eng.ctx.handle, C.c_void_p(dev_frames[slot].data_ptr()), n, self.W, self.H,
                        C.c_void_p(dev_wire[slot].data_ptr()), eng.stream()), "aur_rgb24_to_" + self.wire_format)
                rendered = torch.cuda.Event()
                rendered.record(cur)
                copy_stream.wait_event(rendered)
                _lib.check(eng.lib.aur_readback_async(eng.ctx.handle, C.c_void_p(dev_wire[slot].data_ptr()),
                                                      n * self.frame_bytes, host[slot].ptr,
                                                      C.c_void_p(copy_stream.cuda_stream)), "aur_readback_async")
                ev = torch.cuda.Event()
                ev.record(copy_stream)
                copied[slot] = ev
                q.put((slot, n, ev, done))
                done += n
                i += 1
        finally:
            q.put(None)
            th.join()
            torch.cuda.synchronize(eng.device)
            if own_ring:
                self.close()
        if err and not isinstance(err[0], BrokenPipeError):
            raise err[0]
        return done
