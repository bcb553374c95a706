"""Drop-in for the reference's ``export/exporter.py`` on MI355X.

Keeps the surface the UI and ``export.worker`` use (export/exporter.py):

* ``Exporter(audio_path, width, height, fps, mode, color, sensitivity, prefer_hevc, view_state,
  gpu_rendering, gpu_device)`` with ``render_to_file(out_path, progress_cb)`` and
  ``_run_pipe(renderer, cmd, progress_cb)``  (:494-567);
* a renderer with the duck type ``.duration`` + ``.render_time(t) -> uint8[h, w, 3]``  (:98-252);
* ``build_ffmpeg_cmd(...) -> (argv, "rgb24")`` and ``list_gpu_export_devices()``  (:368-491).

The frame source is ``HipOffscreenRenderer``: the track is uploaded to HBM once, analysed in one fused
launch, rasterised in batches and read back through a pinned ring (``aurora_amd.pipeline``).  PySide6 and
soundfile are optional; there is no CPU rendering fallback.

``gpu_device`` keeps its reference meaning (an FFmpeg *encoder* id).  The compute device is selected with the
new key ``hip_device`` (int) in ``view_state``; time sharding with ``hip_frame_range`` = [begin, end) or
``hip_shard`` = [rank, world] (``aurora_amd.export.sharded`` starts one such process per GPU).
"""
from __future__ import annotations

import ctypes as C
import errno
import logging
import os
import re
import platform
import shutil
import subprocess
import threading
import wave
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

from ..engine import FillConfig, GlowConfig, RadialConfig, ShadowConfig, get_engine, mode_kind, SUPPORTED_KINDS
from ..images import BackgroundConfig
from ..pipeline import FramePipeline

logger = logging.getLogger(__name__)


# ----------------------------------------------------------------------------- audio ingest

def _read_wav_stdlib(path):
    """PCM WAV -> (float32 [n, C] in [-1, 1), samplerate) using the stdlib (soundfile-free fallback)."""
    with wave.open(path, "rb") as w:
        ch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif sw == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        a = v.astype(np.float32) / 8388608.0
    else:
        raise RuntimeError(f"unsupported WAV sample width {sw}")
    return a.reshape(-1, ch), sr


def load_audio(path):
    """Whole file as float32 [n, C] + samplerate: soundfile if installed (exporter.py:110), else float/PCM WAV
    through scipy/stdlib, else FFmpeg decode to mono 48 kHz (the reference's _FFmpegDecodedAudio, :46-81)."""
    try:
        import soundfile as sf
        data, sr = sf.read(path, dtype="float32", always_2d=True)
        return np.ascontiguousarray(data), int(sr)
    except ImportError:
        pass
    except Exception:
        return _decode_with_ffmpeg(path)
    try:
        from scipy.io import wavfile
        sr, data = wavfile.read(path)
        if data.dtype == np.float32:
            a = data
        elif data.dtype == np.int16:
            a = data.astype(np.float32) / 32768.0
        elif data.dtype == np.int32:
            a = data.astype(np.float32) / 2147483648.0
        elif data.dtype == np.uint8:
            a = (data.astype(np.float32) - 128.0) / 128.0
        else:
            a = data.astype(np.float32)
        if a.ndim == 1:
            a = a[:, None]
        return np.ascontiguousarray(a), int(sr)
    except Exception:
        pass
    try:
        return _read_wav_stdlib(path)
    except Exception:
        return _decode_with_ffmpeg(path)


def _decode_with_ffmpeg(path, target_sr=48000):
    ffmpeg = shutil.which("ffmpeg")
    if ffmpeg is None:
        raise RuntimeError("ffmpeg not found for decoding non-WAV formats.")
    cmd = [ffmpeg, "-v", "error", "-i", path, "-f", "f32le", "-acodec", "pcm_f32le", "-ac", "1",
           "-ar", str(int(target_sr)), "-"]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if proc.returncode != 0:
        raise RuntimeError(f"ffmpeg decode failed: {proc.stderr.decode('utf-8', 'ignore')}")
    return np.frombuffer(proc.stdout, dtype=np.float32).reshape(-1, 1).copy(), int(target_sr)


# ----------------------------------------------------------------------------- renderer

class HipOffscreenRenderer:
    """Frame source with the reference renderer's constructor and duck type (exporter.py:98-122, :246-252)."""

    def __init__(self, width, height, mode, color, sensitivity, fps, audio_path, **view_state):
        self.width = int(width)
        self.height = int(height)
        self.mode = str(mode)
        self.color = color
        self.sensitivity = float(sensitivity or 1.0)
        self.fps = int(fps)
        self.audio_path = audio_path
        self.view_state = dict(view_state)

        # sizes the kernels take (the reference UI offers 64..3840 x 64..2160, any integer): checked before the track is
        # loaded, analysed or FFmpeg is started
        if not (64 <= self.width <= 16384 and 16 <= self.height <= 16384):
            raise ValueError(f"frame size {self.width}x{self.height} outside [64..16384]x[16..16384]")
        kind = mode_kind(self.mode)
        if kind not in SUPPORTED_KINDS:
            raise NotImplementedError(f"mode {self.mode!r} is not implemented on the HIP export path yet")
        radial = RadialConfig.from_view_state(view_state, self.sensitivity) if kind == "spectrum_radial" else None

        device = int(view_state.get("hip_device", 0) or 0)
        self._eng = get_engine(device)
        if isinstance(audio_path, (tuple, list)) and len(audio_path) == 2:      # (array, samplerate): tests/bench
            audio, sr = audio_path
        else:
            audio, sr = load_audio(audio_path)
        self._track = (self._eng.upload_track(audio, sr, self.fps) if isinstance(audio, np.ndarray)
                       else self._eng.make_track(audio, sr, self.fps))
        self.samplerate = int(sr)
        self.duration = float(self._track.n_samples / self.samplerate)           # exporter.py:113
        self.total_frames = self._track.total_frames
        rng = view_state.get("hip_frame_range")
        shard = view_state.get("hip_shard")               # [rank, world]: this process renders one time shard
        if shard and not rng:
            from ..sharding import frame_range
            rng = frame_range(self.total_frames, int(shard[1]), int(shard[0]))
        f0, f1 = (int(rng[0]), int(rng[1])) if rng else (0, self.total_frames)
        self.frame_begin, self.frame_end = max(0, f0), min(self.total_frames, f1)
        self._pipe = FramePipeline(self._eng, self._track, self.mode, self.width, self.height, self.sensitivity,
                                   self.frame_begin, self.frame_end,
                                   batch_frames=view_state.get("hip_batch_frames"), radial=radial,
                                   rotation_deg=float(view_state.get("radial_rotation_deg", 0.0)),
                                   wire_format=str(view_state.get("hip_wire_format", "rgb24")),
                                   fill=FillConfig.from_view_state(view_state),
                                   shadow=ShadowConfig.from_view_state(view_state),
                                   glow=GlowConfig.from_view_state(view_state),
                                   background=BackgroundConfig.from_view_state(view_state))
        self.wire_format = self._pipe.wire_format
        self._cache = None          # (first_frame, uint8 [n, H, W, 3] host array)
        self._stepper = None        # state of the call-by-call path (render_time off the frame grid)

    # -- batched fast path used by Exporter._run_pipe ---------------------------------------------------
    def stream_to(self, sink, progress=None):
        return self._pipe.run(sink, progress)

    # -- reference duck type ----------------------------------------------------------------------------
    def render_frame_index(self, i: int) -> np.ndarray:
        i = int(i)
        if not (self.frame_begin <= i < self.frame_end):
            raise IndexError(f"frame {i} outside [{self.frame_begin}, {self.frame_end})")
        c = self._cache
        if c is None or not (c[0] <= i < c[0] + c[1].shape[0]):
            n = min(self._pipe.batch, self.frame_end - i)
            frames = self._pipe.render_device(first=i - self.frame_begin, count=n)
            c = self._cache = (i, frames.cpu().numpy())
        return c[1][i - c[0]]

    def render_frame(self, t, stepwise=None):
        """Frame at time ``t`` (exporter.py:246-249).

        The reference's renderer is a state machine: every call reads ``hop`` samples at ``int(t * sr)`` (:227-237),
        advances its analyser (previous block, previous magnitudes, flux EMA), its colour EMA and its phase by one step
        and draws.  Its only caller asks for ``t = i / fps``, i = 0, 1, 2, ... (:539), and for those times this renderer
        returns frame ``i`` of that sequence from the batched path without replaying the calls before it.  Any other
        ``t`` (or ``stepwise=True``) is served literally: one step of a per-renderer state machine that starts from the
        zero state at its first such call -- position ``int(t * sr)``, analysis through the same fused kernel over the
        positions of the last three such calls, scan with the carried scalar state, ring EMA with the carried ring."""
        x = float(t) * self.fps
        i = int(round(x))
        on_grid = abs(x - i) <= 1e-9 * max(1.0, abs(x))
        if stepwise is None:
            stepwise = not on_grid
        if not stepwise:
            return self.render_frame_index(i)
        return self._step(float(t))

    def render_time(self, t, stepwise=None):
        return self.render_frame(t, stepwise)

    def _step(self, t):
        from ..engine import Track
        eng, tr, torch = self._eng, self._track, self._eng.torch
        st = self._stepper
        if st is None:
            st = self._stepper = {"pos": [], "state": None, "ring": None}
        pos = int(max(0, min(tr.n_samples - 1, t * self.samplerate)))          # exporter.py:230
        hist = (st["pos"] + [pos])[-3:]        # frame k needs block k, block k-1 and |X| of frame k-1 (blocks k-1, k-2)
        st["pos"] = hist
        table = np.asarray(hist, dtype=np.int64)
        view = Track(tr.audio, torch.from_numpy(table).to(eng.device), table, tr.n_samples, tr.channels, tr.sample_rate,
                     tr.fps, tr.hop, len(hist))
        pipe = self._pipe
        an = eng.analyze(view, len(hist) - 1, 1, want_samples=pipe.kind.startswith("waveform"))
        sc = eng.scan(an, pipe.params, st["state"])
        st["state"] = tuple(float(v) for v in sc.state.cpu().numpy())
        prep = None
        if pipe.radial is not None:
            prep = eng.radial_prepare(an, pipe.radial, st["ring"])
            st["ring"] = prep.state.clone() if pipe.radial.radial_temporal_alpha > 0.0 else None
        return pipe.render_one(an, sc, prep)[0].cpu().numpy()


# ----------------------------------------------------------------------------- ffmpeg utilities

def _ffmpeg_candidates():
    """Where an FFmpeg binary may come from, in the reference's order of preference (exporter.py:319-341): the two
    environment variables, then whatever the imageio_ffmpeg package ships."""
    for var in ("AURORA_FFMPEG", "IMAGEIO_FFMPEG_EXE"):
        yield os.environ.get(var)
    try:
        import imageio_ffmpeg
        yield imageio_ffmpeg.get_ffmpeg_exe()
    except Exception:
        return


def _ffmpeg_bin() -> str:
    for cand in _ffmpeg_candidates():
        if cand and os.path.exists(cand):
            return cand
    on_path = shutil.which("ffmpeg")
    if on_path is None:
        raise RuntimeError("FFmpeg not found. Install ffmpeg or pip install imageio-ffmpeg, "
                           "or set AURORA_FFMPEG/IMAGEIO_FFMPEG_EXE.")
    return on_path


# a video encoder line of `ffmpeg -encoders`: " V....D libx264   H.264 / AVC ..." -> the second column
_VIDEO_ENCODER_LINE = re.compile(r"^ V\S*[ \t]+(\S+)", re.MULTILINE)


def _ffmpeg_encoders(ffmpeg: str) -> set:
    """Names of the video encoders this FFmpeg offers; empty when it cannot be asked (exporter.py:344-360)."""
    try:
        proc = subprocess.run([ffmpeg, "-hide_banner", "-encoders"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True, errors="ignore")
    except Exception:
        return set()
    return set(_VIDEO_ENCODER_LINE.findall(proc.stdout)) if proc.returncode == 0 else set()


def _ensure_ext(path: str, default_ext: str) -> str:
    return path if os.path.splitext(path)[1] else os.path.splitext(path)[0] + default_ext


# encoder family -> (h264 name, hevc name, extra video flags)
_HW = {
    "videotoolbox": ("h264_videotoolbox", "hevc_videotoolbox", ["-b:v", "12M", "-maxrate", "20M", "-bufsize", "40M"]),
    "nvenc": ("h264_nvenc", "hevc_nvenc", ["-preset", "p5", "-rc", "vbr", "-cq", "19", "-b:v", "0", "-maxrate", "0"]),
    "vaapi": ("h264_vaapi", "hevc_vaapi", ["-b:v", "12M", "-maxrate", "20M", "-bufsize", "40M"]),
    "qsv": ("h264_qsv", "hevc_qsv", ["-global_quality", "23"]),
    "amf": ("h264_amf", "hevc_amf", ["-quality", "balanced"]),
}


def list_gpu_export_devices() -> List[Dict[str, str]]:
    """Hardware video ENCODERS FFmpeg offers (UI dropdown, exporter.py:368-410)."""
    enc = _ffmpeg_encoders(_ffmpeg_bin())
    system = platform.system()

    def has(fam):
        return _HW[fam][0] in enc or _HW[fam][1] in enc

    if system == "Darwin":
        return [{"id": "videotoolbox", "label": "Apple VideoToolbox"}] if has("videotoolbox") else []
    out: List[Dict[str, str]] = []
    if has("nvenc"):
        lines: List[str] = []
        smi = shutil.which("nvidia-smi")
        if smi:
            try:
                txt = subprocess.check_output([smi, "-L"], text=True, errors="ignore")
                lines = [ln.strip() for ln in txt.splitlines() if ln.strip().startswith("GPU ")]
            except Exception:
                lines = []
        if lines:
            out += [{"id": f"nvenc:{k}", "label": f"NVIDIA NVENC ({ln})"} for k, ln in enumerate(lines)]
        else:
            out.append({"id": "nvenc:0", "label": "NVIDIA NVENC"})
    if has("amf"):
        out.append({"id": "amf:auto", "label": "AMD AMF"})
    if has("qsv"):
        out.append({"id": "qsv:auto", "label": "Intel Quick Sync (QSV)"})
    if has("vaapi") and system == "Linux":
        try:
            for node in sorted(Path("/dev/dri").glob("renderD*")):
                out.append({"id": f"vaapi:{node}", "label": f"VAAPI ({node.name})"})
        except Exception:
            pass
    return out


def _encoder_plan(prefer_hevc: bool, gpu_device: Optional[str]):
    """(ffmpeg, device args, filter, codec, codec flags, output pix_fmt) for an FFmpeg *encoder* id
    (exporter.py:422-483): hardware family when its encoder exists, else libx264."""
    ffmpeg = _ffmpeg_bin()
    enc = _ffmpeg_encoders(ffmpeg)
    system = platform.system()

    dev = (gpu_device or "").strip()
    if dev.lower() in {"", "0", "false", "off", "no", "cpu"}:
        dev = ""
    elif dev.lower() == "auto":
        choices = list_gpu_export_devices()
        dev = choices[0]["id"] if choices else ""

    fam = dev.split(":", 1)[0] if dev else ""
    arg = dev.split(":", 1)[1] if ":" in dev else ""
    if fam == "videotoolbox":            # exact id, macOS only; a missing encoder is an error, not a fallback (:446-448)
        usable = dev == "videotoolbox" and system == "Darwin"
    else:                                # "<family>:<arg>" and at least one encoder of the family, else libx264 (:450-476)
        usable = (fam in _HW and ":" in dev and (_HW[fam][0] in enc or _HW[fam][1] in enc)
                  and (fam != "vaapi" or system == "Linux"))
    pix_out, vfilter, dev_args = "yuv420p", None, []
    if usable:
        h264, hevc, flags = _HW[fam]
        codec = hevc if (prefer_hevc and hevc in enc) else h264
        if codec not in enc:
            raise RuntimeError(f"Hardware encoder not available: {codec}")
        vflags = list(flags)
        if fam == "nvenc":
            try:
                idx = int(arg)
            except Exception:
                idx = 0
            vflags = ["-gpu", str(idx)] + vflags
        elif fam == "vaapi":
            dev_args = ["-vaapi_device", arg]
            vfilter, pix_out = "format=nv12,hwupload", "nv12"
    else:
        codec, vflags = "libx264", ["-preset", "veryfast", "-crf", "20"]
    return ffmpeg, dev_args, vfilter, codec, vflags, pix_out


def _check_pix_fmt(input_pix_fmt):
    if input_pix_fmt not in ("rgb24", "yuv420p", "nv12"):
        raise ValueError(f"input_pix_fmt {input_pix_fmt!r}")


def build_ffmpeg_cmd(width: int, height: int, fps: int, out_path: str, audio_path: str,
                     prefer_hevc: bool = False, gpu_device: Optional[str] = None,
                     input_pix_fmt: str = "rgb24") -> Tuple[List[str], str]:
    """FFmpeg argv for rawvideo WxH@fps on stdin + the audio file (exporter.py:413-491).  ``input_pix_fmt`` is the
    reference's "rgb24" unless the renderer was asked for the yuv420p or nv12 wire format (view_state["hip_wire_format"];
    nv12 is what the VAAPI chain "format=nv12,hwupload" converts to, :464-466, so that filter then passes frames through)."""
    _check_pix_fmt(input_pix_fmt)
    out_path = _ensure_ext(out_path, ".mp4")
    ffmpeg, dev_args, vfilter, codec, vflags, pix_out = _encoder_plan(prefer_hevc, gpu_device)
    argv: List[str] = [ffmpeg, "-y", "-f", "rawvideo", "-pix_fmt", input_pix_fmt, "-s", f"{width}x{height}",
                       "-r", str(int(fps)), "-i", "pipe:0", "-i", audio_path]
    argv += dev_args
    if vfilter:
        argv += ["-vf", vfilter]
    argv += ["-c:v", codec, *vflags, "-pix_fmt", pix_out, "-c:a", "aac", "-b:a", "192k", "-shortest", out_path]
    return argv, input_pix_fmt


def build_ffmpeg_segment_cmd(width: int, height: int, fps: int, out_path: str, prefer_hevc: bool = False,
                             gpu_device: Optional[str] = None, input_pix_fmt: str = "rgb24") -> Tuple[List[str], str]:
    """The same encode for one time shard of a multi-GPU export (aurora_amd.export.sharded): video only (``-an``); the
    supervisor joins the segments and muxes the audio.  Not in the reference, which encodes in one process."""
    _check_pix_fmt(input_pix_fmt)
    out_path = _ensure_ext(out_path, ".mp4")
    ffmpeg, dev_args, vfilter, codec, vflags, pix_out = _encoder_plan(prefer_hevc, gpu_device)
    argv: List[str] = [ffmpeg, "-y", "-f", "rawvideo", "-pix_fmt", input_pix_fmt, "-s", f"{width}x{height}",
                       "-r", str(int(fps)), "-i", "pipe:0"]
    argv += dev_args
    if vfilter:
        argv += ["-vf", vfilter]
    argv += ["-c:v", codec, *vflags, "-pix_fmt", pix_out, "-an", out_path]
    return argv, input_pix_fmt


# ----------------------------------------------------------------------------- exporter

class _PipeSink:
    """The write end of the pipe into FFmpeg (``proc.stdin.write(frame.tobytes())``, exporter.py:542).

    The pipeline hands over whole ring slots (about 1 GiB of frames in page-locked memory).  A plain ``write`` makes the
    kernel allocate a pipe page and copy into it for every 4 KiB, one thread, before the reader copies it out again
    (3.6 GB/s = 145 4K frames/s measured on the MI355X box, ``profiles/r03_pipe_*``).  ``vmsplice`` puts user pages into the
    pipe instead, which leaves the reader's copy as the only single-threaded one (26 GB/s against the same reader).

    * **direct**: ordinary (pageable) caller memory is spliced as it is.  Spliced pages stay referenced by the pipe until
      the reader has taken them, and the caller reuses its buffer as soon as ``write`` returns, so the last ``capacity``
      bytes of every buffer go through an ordinary ``write``: a pipe is first-in first-out and holds at most ``capacity``
      bytes in pages of 4 KiB, so by the time that copy has fitted, every spliced page before it has left the pipe.
    * **staged**: the ring itself cannot be spliced -- ``hipHostMalloc`` memory is a device-file mapping (``vmsplice``
      answers EFAULT) and this pool refuses ``hipHostRegister`` of anonymous memory (``profiles/r03_hostreg_probe.txt``) --
      so chunks of it (32 MiB) are copied by a few threads into three page-aligned staging buffers of the sink's own and spliced
      from there; a staging buffer is overwritten only after a whole later chunk (at least ``capacity`` bytes) has been
      spliced behind it, which by the same first-in first-out argument means its pages have left the pipe.

    Either way ``write`` means what it always meant: on return the pipe refers to nothing of the caller's memory.
    One assumption about the far end: it takes bytes out of the pipe by READING them (FFmpeg does: ``read`` into its own
    packet buffers).  A consumer that moves the pipe's pages onward without copying (``splice`` / ``tee`` into a socket or
    another pipe) would keep references to staging pages after they have "left the pipe"; for such a consumer set
    ``AUR_PIPE_SPLICE=0``.
    Buffers under ``4 * capacity`` and hosts without ``vmsplice`` take the plain path.  ``AUR_PIPE_SPLICE=0`` turns
    splicing off, ``AUR_PIPE_COPY_THREADS`` sets the staging copy threads (default 3), ``AUR_PIPE_CHUNK_MB`` the staging chunk (default 32;
    sweeps of both: ``profiles/r03_pipe_tune.txt``)."""

    CHUNK = 32 << 20

    def __init__(self, stdin, splice=None, staged=None):
        self._stdin = stdin
        try:                                      # tuning knob (tools/pipe_shards.py): staging chunk in MiB
            self.CHUNK = max(4, int(os.environ.get("AUR_PIPE_CHUNK_MB", "32"))) << 20
        except ValueError:
            pass
        self.spliced_bytes = 0
        self.copied_bytes = 0
        self.refused_errno = None         # errno of the vmsplice call that sent this sink to the staged path
        self._fd = None
        self._cap = 0
        self._staged = bool(staged)       # None / False: found out from the first refusal
        self._stage = None                # (buffers, thread pool), made on first use
        self._turn = 0
        if splice is None:
            splice = os.environ.get("AUR_PIPE_SPLICE", "1") != "0"
        if splice:
            try:
                import fcntl
                self._fd = stdin.fileno()
                self._cap = int(fcntl.fcntl(self._fd, getattr(fcntl, "F_GETPIPE_SZ", 1032)))
                self._vmsplice = _libc_vmsplice()
            except Exception:
                self._fd = None
            if self._fd is not None and (self._vmsplice is None or self._cap <= 0 or 4 * self._cap > self.CHUNK):
                self._fd = None

    def write(self, buf):
        view = memoryview(buf)
        n = view.nbytes
        if self._fd is None or n < 4 * self._cap:
            self.copied_bytes += n
            self._stdin.write(buf)
            return
        self._stdin.flush()                                  # nothing of an earlier write may overtake these bytes
        flat = np.frombuffer(view, dtype=np.uint8)
        off = 0
        if not self._staged:
            head = n - self._cap
            off = self._splice(flat.ctypes.data, head)
            if off < head:                                   # refused before a byte went in: not spliceable memory
                self._staged = True
        if self._staged and off == 0:
            self._write_staged(flat)
            return
        self.spliced_bytes += off
        self.copied_bytes += n - off
        rest = view.cast("B")[off:]
        while len(rest):
            rest = rest[os.write(self._fd, rest):]

    # ---- staged path -------------------------------------------------------------------------------------------------
    def _staging(self):
        if self._stage is None:
            import mmap
            from concurrent.futures import ThreadPoolExecutor
            size = self.CHUNK + self._cap
            maps = [mmap.mmap(-1, size) for _ in range(3)]                       # page-aligned, the sink's own
            bufs = [np.frombuffer(m, dtype=np.uint8) for m in maps]
            threads = max(1, int(os.environ.get("AUR_PIPE_COPY_THREADS", "3") or 3))    # profiles/r03_pipe_tune.txt
            self._stage = (bufs, ThreadPoolExecutor(threads, thread_name_prefix="aurora-pipe-stage"), threads, maps)
        return self._stage

    def _copy_into(self, slot, src):
        """Starts the copy of ``src`` into staging buffer ``slot`` on the pool; returns the futures."""
        bufs, pool, threads, _ = self._staging()
        n = src.size
        step = ((n + threads - 1) // threads + 4095) // 4096 * 4096          # whole pages per thread
        return [pool.submit(np.copyto, bufs[slot][o:min(n, o + step)], src[o:min(n, o + step)])
                for o in range(0, n, step)]

    def _write_staged(self, flat):
        n, chunk = flat.size, self.CHUNK
        cuts = list(range(0, n, chunk))
        if len(cuts) > 1 and n - cuts[-1] < self._cap:       # every chunk holds at least `capacity` bytes
            cuts.pop()
        cuts.append(n)
        bufs = self._staging()[0]
        pending = self._copy_into(self._turn % 3, flat[cuts[0]:cuts[1]])
        try:
            for k in range(len(cuts) - 1):
                for f in pending:
                    f.result()
                slot = self._turn % 3
                self._turn += 1
                pending = []
                if k + 2 < len(cuts):
                    # buffer (turn % 3) was last spliced two chunks ago; the chunk spliced after it has gone in whole
                    pending = self._copy_into(self._turn % 3, flat[cuts[k + 1]:cuts[k + 2]])
                size = cuts[k + 1] - cuts[k]
                if self._splice(bufs[slot].ctypes.data, size) != size:
                    raise OSError(self.refused_errno or errno.EIO, "vmsplice refused the sink's own staging buffer")
                self.spliced_bytes += size
        finally:
            for f in pending:                                # never leave a copy running into the caller's buffer
                f.cancel()
            for f in pending:
                if not f.cancelled():
                    try:
                        f.result()
                    except Exception:
                        pass

    def close(self):
        if self._stage is not None:
            self._stage[1].shutdown(wait=True)
            self._stage = None

    def _splice(self, addr, nbytes):
        """vmsplice ``nbytes`` at ``addr``; returns how many went in (less than asked only if the kernel refused the
        memory before anything was sent)."""
        off = 0
        iov = _IoVec()
        while off < nbytes:
            iov.iov_base, iov.iov_len = addr + off, nbytes - off
            k = self._vmsplice(self._fd, C.byref(iov), 1, 0)
            if k < 0:
                e = C.get_errno()
                if e == errno.EINTR:
                    continue
                if e == errno.EPIPE:
                    raise BrokenPipeError(e, os.strerror(e))
                if off == 0 and e in (errno.EFAULT, errno.EINVAL, errno.ENOSYS, errno.EBADF, errno.ENOMEM):
                    self.refused_errno = e
                    return 0
                raise OSError(e, os.strerror(e))
            off += k
        return off


class _IoVec(C.Structure):
    _fields_ = [("iov_base", C.c_void_p), ("iov_len", C.c_size_t)]


def _libc_vmsplice():
    """``ssize_t vmsplice(int fd, const struct iovec*, size_t nr_segs, unsigned flags)`` of the C library, or None."""
    try:
        fn = C.CDLL(None, use_errno=True).vmsplice
    except (OSError, AttributeError):
        return None
    fn.restype = C.c_ssize_t
    fn.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_uint]
    return fn


def _grow_pipe(stream, want=1 << 20):
    """A 4K rgb24 frame is 24.9 MB and Linux pipes default to 64 KB, i.e. ~380 writer/reader hand-overs per frame:
    ask for the largest pipe an unprivileged process may have (/proc/sys/fs/pipe-max-size, 1 MiB by default).  Best
    effort: any failure leaves the default pipe in place."""
    try:
        import fcntl
        limit = want
        try:
            with open("/proc/sys/fs/pipe-max-size") as fh:
                limit = min(want, int(fh.read().strip()))
        except (OSError, ValueError):
            pass
        fcntl.fcntl(stream.fileno(), getattr(fcntl, "F_SETPIPE_SZ", 1031), limit)
        return limit
    except Exception:
        return 0


class _TailReader:
    """Reads a child's pipe to its end on a daemon thread, keeping the last ``keep`` bytes."""

    def __init__(self, stream, keep):
        self._stream, self._keep, self._chunks, self._size = stream, int(keep), [], 0
        self._thread = None
        if stream is not None:
            self._thread = threading.Thread(target=self._drain, name="aurora-ffmpeg-drain", daemon=True)
            self._thread.start()

    def _drain(self):
        try:
            while True:
                chunk = self._stream.read1(1 << 16) if hasattr(self._stream, "read1") else self._stream.read(1 << 16)
                if not chunk:
                    return
                if self._keep:
                    self._chunks.append(chunk)
                    self._size += len(chunk)
                    while self._size - len(self._chunks[0]) >= self._keep:
                        self._size -= len(self._chunks.pop(0))
        except Exception:
            pass

    def join(self):
        if self._thread is not None:
            self._thread.join()

    def text(self):
        self.join()
        return b"".join(self._chunks)[-self._keep:].decode("utf-8", errors="ignore") if self._keep else ""


class Exporter:
    def __init__(self, audio_path, width=1920, height=1080, fps=60, mode="Waveform - Linear",
                 color=(0.2, 0.8, 1.0), sensitivity=1.0, prefer_hevc=False, view_state=None,
                 gpu_rendering=False, gpu_device=None):
        self.audio_path = audio_path
        self.width = int(width)
        self.height = int(height)
        self.fps = int(fps)
        self.mode = str(mode)
        self.color = tuple(color)
        self.sensitivity = float(sensitivity)
        self.prefer_hevc = bool(prefer_hevc)
        self.view_state = view_state or {}
        self.gpu_rendering = bool(gpu_rendering)
        self.gpu_device = gpu_device or ("auto" if self.gpu_rendering else "")
        self.last_stats = None            # filled by every export (see _pump)

    def make_renderer(self):
        return HipOffscreenRenderer(self.width, self.height, self.mode, self.color, self.sensitivity, self.fps,
                                    self.audio_path, **self.view_state)

    def render_to_file(self, out_path, progress_cb=None):
        renderer = self.make_renderer()
        logger.info("Export using HIP renderer on MI355X (device %s)", self.view_state.get("hip_device", 0))
        wire = getattr(renderer, "wire_format", "rgb24")
        if self.view_state.get("hip_segment") and renderer.frame_end <= renderer.frame_begin:
            # a time shard with no frames (more GPUs than frames): FFmpeg fails on an empty rawvideo input, and the
            # supervisor's join skips segments that do not exist
            if progress_cb:
                progress_cb(100)
            return None
        if self.view_state.get("hip_segment"):           # one shard of a multi-GPU export: video-only segment
            cmd, _ = build_ffmpeg_segment_cmd(self.width, self.height, self.fps, out_path, prefer_hevc=self.prefer_hevc,
                                              gpu_device=self.gpu_device, input_pix_fmt=wire)
        else:
            cmd, _ = build_ffmpeg_cmd(self.width, self.height, self.fps, out_path, self.audio_path,
                                      prefer_hevc=self.prefer_hevc, gpu_device=self.gpu_device, input_pix_fmt=wire)
        return self._run_pipe(renderer, cmd, progress_cb=progress_cb)

    def render_to_sink(self, sink, progress_cb=None):
        """Same frame stream as render_to_file, into any object with write(buffer) (tests, benchmarks, raw files)."""
        renderer = self.make_renderer()
        return self._pump(renderer, sink, progress_cb)

    def _pump(self, renderer, sink, progress_cb):
        """Frames of ``renderer`` into ``sink`` in order; fills ``self.last_stats`` (frames, seconds, frames/s, bytes on
        the wire and, for a pipe sink, how they went in) -- the per-export figures a deployment logs
        (``view_state["hip_stats_path"]`` makes the worker write them as JSON)."""
        import time
        t0 = time.perf_counter()
        n = self._pump_frames(renderer, sink, progress_cb)
        dt = time.perf_counter() - t0
        per_frame = int(getattr(getattr(renderer, "_pipe", None), "frame_bytes", self.width * self.height * 3))
        st = {"frames": int(n), "seconds": round(dt, 6), "fps": round(n / dt, 2) if dt > 0 else None,
              "wire_format": getattr(renderer, "wire_format", "rgb24"), "bytes_per_frame": per_frame,
              "wire_gbs": round(n * per_frame / dt / 1e9, 3) if dt > 0 else None,
              "frame_range": [int(getattr(renderer, "frame_begin", 0)), int(getattr(renderer, "frame_begin", 0)) + int(n)],
              "device": int(self.view_state.get("hip_device", 0) or 0)}
        if isinstance(sink, _PipeSink):
            st["pipe"] = {"spliced_bytes": int(sink.spliced_bytes), "copied_bytes": int(sink.copied_bytes),
                          "path": ("write" if sink.spliced_bytes == 0 else "staged vmsplice" if sink._staged else "vmsplice")}
        self.last_stats = st
        return n

    def _pump_frames(self, renderer, sink, progress_cb):
        total_frames = int(np.ceil(renderer.duration * self.fps)) if hasattr(renderer, "duration") else 0   # :536
        step = max(1, int(self.fps / 2))                                                                   # :545

        def _tick(i):
            if progress_cb and (i % step == 0):
                try:
                    progress_cb(int(i * 100 / total_frames) if total_frames else 0)
                except Exception:
                    pass

        if hasattr(renderer, "stream_to"):
            seen = [getattr(renderer, "frame_begin", 0)]

            def on_batch(done, _n):
                upto = getattr(renderer, "frame_begin", 0) + done
                for i in range(seen[0], upto):
                    _tick(i)
                seen[0] = upto

            return renderer.stream_to(sink, on_batch)
        written = 0
        for i in range(total_frames):                      # generic duck-typed renderer: the reference loop
            frame = np.ascontiguousarray(renderer.render_time(i / self.fps))
            try:
                sink.write(memoryview(frame).cast("B"))
            except BrokenPipeError:
                break
            written += 1
            _tick(i)
        return written

    def _run_pipe(self, renderer, cmd, progress_cb=None):
        proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                bufsize=10 ** 7)
        # FFmpeg prints statistics while it encodes: both pipes are drained while the frames are written, or the child
        # blocks on a full pipe a few minutes in and the writer behind it for good.  The tail of stderr is kept for the
        # error message.
        _grow_pipe(proc.stdin)
        tail = _TailReader(proc.stderr, keep=1 << 16)
        sinkhole = _TailReader(proc.stdout, keep=0)
        sink = _PipeSink(proc.stdin)
        try:
            self._pump(renderer, sink, progress_cb)
        finally:
            sink.close()
            try:
                if proc.stdin:
                    proc.stdin.close()
            except Exception:
                pass
        code = proc.wait()
        stderr = tail.text()
        sinkhole.join()
        if progress_cb:
            try:
                progress_cb(100)
            except Exception:
                pass
        if code != 0:
            raise RuntimeError(f"ffmpeg failed (exit {code}):\n{stderr}")
