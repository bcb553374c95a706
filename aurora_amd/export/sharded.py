"""Multi-GPU export driver: one child process per GPU, contiguous frame ranges, segments joined in frame order.

    python -m aurora_amd.export.sharded cfg.json --gpus N              # one track, time-sharded over N GPUs
    python -m aurora_amd.export.sharded a.json b.json ... --gpus N     # many tracks, dealt over N GPUs

The reference exports one frame after another in one process (``export/exporter.py:531-549``).  Here the process that
is started (this module, or ``export.worker`` with ``view_state["hip_gpus"] = N``) only supervises: it never touches a
GPU.  For ONE track it starts N children of ``aurora_amd.export.worker`` before anything else; child r is told
``view_state["hip_shard"] = [r, N]`` and ``view_state["hip_device"]``, renders frames
``sharding.frame_range(total, N, r)`` (analysis halo inside the STFT kernel, carried smoothing state from an
analysis-only pre-pass, so its bytes equal the same frames of an unsharded export) and writes its own segment:

* ``hip_raw_out`` set: a raw segment file ``<hip_raw_out>.seg<r>``; the supervisor appends them in rank order;
* otherwise: its own FFmpeg encoding a video-only segment ``<out>.seg<r><ext>`` (the encoder arguments of
  ``build_ffmpeg_cmd``, no audio input); the supervisor joins them with FFmpeg's concat demuxer and muxes the audio
  (``-c:v copy``, the reference's ``-c:a aac -b:a 192k -shortest``).

There is no GPU-to-GPU traffic and no collective: the "final gather" of the north star is this ordered join on the
host.  For MANY tracks (BASELINE.json configs[4]) whole tracks are dealt round-robin
(``sharding.track_assignment``): GPU g exports tracks g, g + N, ... one after another, each in a fresh worker process.

Progress and errors use the worker's protocol on stdout (JSON lines, exit 0 / 1 / 2), so the UI can launch this
instead of the worker.
"""
from __future__ import annotations

import argparse
import copy
import hashlib
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import threading
import time
from pathlib import Path

from .. import sharding


def _emit(msg: dict) -> None:
    try:
        print(json.dumps(msg), file=sys.stdout, flush=True)
    except Exception:
        pass


def default_worker_cmd():
    return [sys.executable, "-m", "aurora_amd.export.worker"]


def _repo_root() -> str:
    return str(Path(__file__).resolve().parents[2])


def _child_env():
    env = dict(os.environ)
    root = _repo_root()
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def segment_path(path: str, rank: int) -> str:
    """``out.mp4`` -> ``out.seg003.mp4``; a path without extension gets the suffix appended."""
    stem, ext = os.path.splitext(path)
    return f"{stem}.seg{rank:03d}{ext}"


def shard_configs(cfg: dict, world: int, devices=None) -> list:
    """Worker configs of the ``world`` time shards of one track (rank order = frame order)."""
    world = int(world)
    if world < 1:
        raise ValueError(f"--gpus {world}")
    devices = list(devices) if devices else list(range(world))
    out = []
    for r in range(world):
        c = copy.deepcopy(cfg)
        vs = dict(c.get("view_state") or {})
        vs.pop("hip_gpus", None)
        vs.pop("hip_devices", None)
        vs["hip_shard"] = [r, world]
        vs["hip_device"] = int(devices[r % len(devices)])
        if vs.get("hip_stats_path"):                       # every shard reports its own figures
            vs["hip_stats_path"] = segment_path(str(vs["hip_stats_path"]), r)
        if c.get("hip_raw_out"):
            c["hip_raw_out"] = segment_path(str(c["hip_raw_out"]), r)
        else:
            vs["hip_segment"] = True                       # video-only segment: the join muxes the audio
            c["out_path"] = segment_path(_with_ext(str(c["out_path"])), r)
        c["view_state"] = vs
        out.append(c)
    return out


def _with_ext(path: str, default_ext: str = ".mp4") -> str:
    return path if os.path.splitext(path)[1] else path + default_ext


class _Child:
    """One worker process: its stdout protocol is parsed on a thread, stderr is kept for the error message."""

    def __init__(self, cfg: dict, tmpdir: str, tag: str, worker_cmd, on_progress):
        self.tag = tag
        self.cfg_path = os.path.join(tmpdir, f"{tag}.json")
        with open(self.cfg_path, "w", encoding="utf-8") as fh:
            json.dump(cfg, fh)
        self.pct = 0
        self.error = None
        self.done = False
        self.last_sign = time.monotonic()          # when the worker last said anything (stall detection)
        self._on_progress = on_progress
        # its own session = its own process group: stopping a worker also stops the FFmpeg it started
        self.proc = subprocess.Popen(list(worker_cmd) + [self.cfg_path], stdout=subprocess.PIPE,
                                     stderr=subprocess.PIPE, env=_child_env(), cwd=_repo_root(),
                                     start_new_session=True)
        self._err_tail = []
        self._threads = [threading.Thread(target=self._read_stdout, daemon=True),
                         threading.Thread(target=self._read_stderr, daemon=True)]
        for t in self._threads:
            t.start()

    def _read_stdout(self):
        for raw in self.proc.stdout:
            self.last_sign = time.monotonic()
            try:
                msg = json.loads(raw.decode("utf-8", "ignore"))
            except ValueError:
                continue
            kind = msg.get("type")
            if kind == "progress":
                try:
                    self.pct = max(self.pct, int(msg.get("pct", 0)))
                except (TypeError, ValueError):
                    continue                                   # a malformed line must not end the reader thread
                self._on_progress()
            elif kind == "error":
                self.error = str(msg.get("message", "export failed"))
            elif kind == "done":
                self.done = True

    def _read_stderr(self):
        for raw in self.proc.stderr:
            self._err_tail.append(raw)
            del self._err_tail[:-40]

    def wait(self):
        code = self.proc.wait()
        for t in self._threads:
            t.join()
        return code

    def failure(self, code):
        if code == 0 and self.done:
            return None
        tail = b"".join(self._err_tail).decode("utf-8", "ignore").strip()
        return f"{self.tag}: {self.error or 'worker exited with status %d' % code}" + (f"\n{tail}" if tail else "")

    KILL_AFTER_S = 10.0      # grace between SIGTERM and SIGKILL for a worker that does not react (blocked in a call)

    def stop(self):
        """Ends the worker AND what it started (its FFmpeg would otherwise go on writing the segment the supervisor is
        about to remove).  SIGTERM to the process group first; a group that is still there ``KILL_AFTER_S`` seconds later
        gets SIGKILL, so the threads waiting for the worker always return."""
        if self.proc.poll() is None:
            try:
                os.killpg(self.proc.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError, AttributeError):
                self.proc.terminate()
            if not getattr(self, "_kill_timer", None):
                def hard():
                    if self.proc.poll() is None:
                        self.reap_group()
                        try:
                            self.proc.kill()
                        except OSError:
                            pass
                self._kill_timer = threading.Timer(self.KILL_AFTER_S, hard)
                self._kill_timer.daemon = True
                self._kill_timer.start()

    def reap_group(self):
        """After the worker has gone: nothing of its process group may be left behind (an FFmpeg that outlived it)."""
        try:
            os.killpg(self.proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError, AttributeError):
            pass


def _run_children(children, stall_timeout_s=None):
    """Waits for every child; the first failure stops the others.  ``stall_timeout_s``: a worker that has been silent for
    that long (no progress line: a hung GPU, a wedged FFmpeg) counts as failed and is stopped with the rest -- the
    reference has no such guard (a stuck export hangs its UI job for good); off unless asked for.
    Returns the failure messages."""
    failures, lock = [], threading.Lock()
    finished = threading.Event()

    def watchdog():
        while not finished.wait(0.2):
            now = time.monotonic()
            for ch in children:
                if ch.proc.poll() is None and now - ch.last_sign > stall_timeout_s:
                    with lock:
                        failures.append(f"{ch.tag}: no sign of life for {stall_timeout_s:.0f} s, stopped")
                    for other in children:
                        other.stop()
                    return

    if stall_timeout_s:
        threading.Thread(target=watchdog, daemon=True).start()

    def watch(ch):
        msg = ch.failure(ch.wait())
        if msg:
            with lock:
                failures.append(msg)
            for other in children:
                if other is not ch:
                    other.stop()

    threads = [threading.Thread(target=watch, args=(ch,), daemon=True) for ch in children]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    finished.set()
    return failures


def file_digest(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 24), b""):
            h.update(chunk)
    return h.hexdigest()


def join_raw_segments(paths, out_path: str, keep: bool = False):
    """Appends the segment files in the order given (rank order = frame order).  Returns [(path, bytes, sha256)]."""
    report = []
    with open(out_path, "wb") as out:
        for p in paths:
            report.append((p, os.path.getsize(p), file_digest(p)))
            with open(p, "rb") as fh:
                shutil.copyfileobj(fh, out, 1 << 24)
    if not keep:
        for p in paths:
            os.remove(p)
    return report


def build_ffmpeg_concat_cmd(list_path: str, audio_path: str, out_path: str):
    """Join of the encoded segments: concat demuxer, video copied, audio muxed as the reference's single command does
    (``-c:a aac -b:a 192k -shortest``, export/exporter.py:485-490)."""
    from .exporter import _ffmpeg_bin
    return [_ffmpeg_bin(), "-y", "-f", "concat", "-safe", "0", "-i", list_path, "-i", audio_path, "-c:v", "copy",
            "-c:a", "aac", "-b:a", "192k", "-shortest", out_path]


def join_encoded_segments(paths, audio_path: str, out_path: str, keep: bool = False):
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False, encoding="utf-8") as fh:
        for p in paths:
            fh.write("file '%s'\n" % os.path.abspath(p).replace("'", "'\\''"))
        list_path = fh.name
    try:
        proc = subprocess.run(build_ffmpeg_concat_cmd(list_path, audio_path, out_path), stdout=subprocess.DEVNULL,
                              stderr=subprocess.PIPE)
        if proc.returncode != 0:
            raise RuntimeError(f"ffmpeg failed (exit {proc.returncode}):\n{proc.stderr.decode('utf-8', 'ignore')}")
    finally:
        os.remove(list_path)
    if not keep:
        for p in paths:
            os.remove(p)


def export_sharded(cfg: dict, world: int, progress_cb=None, worker_cmd=None, devices=None, keep_segments=False,
                   stall_timeout_s=None):
    """One track over ``world`` GPUs.  Raises RuntimeError with the children's messages on failure.
    ``stall_timeout_s`` (or ``view_state["hip_stall_timeout_s"]``): see ``_run_children``."""
    world = int(world)
    if stall_timeout_s is None:
        stall_timeout_s = (cfg.get("view_state") or {}).get("hip_stall_timeout_s")
    stall_timeout_s = float(stall_timeout_s) if stall_timeout_s else None
    children_cfg = shard_configs(cfg, world, devices)
    last = [-1]
    lock = threading.Lock()
    children = []

    def on_progress():
        # child r reports whole-track percentages inside [100 r / world, 100 (r + 1) / world] (it renders that range)
        if progress_cb is None:
            return
        done = 0.0
        for r, ch in enumerate(children):
            lo, hi = 100.0 * r / world, 100.0 * (r + 1) / world
            done += min(max(ch.pct - lo, 0.0), hi - lo)
        pct = int(done)
        with lock:
            if pct > last[0] and pct < 100:
                last[0] = pct
                try:
                    progress_cb(pct)
                except Exception:
                    pass

    with tempfile.TemporaryDirectory(prefix="aurora_shards_") as tmp:
        for r, c in enumerate(children_cfg):                          # all children start before anything else happens
            children.append(_Child(c, tmp, f"shard{r:03d}", worker_cmd or default_worker_cmd(), on_progress))
        failures = _run_children(children, stall_timeout_s)   # returns when every worker has been waited for
    if failures:
        for ch in children:
            ch.reap_group()
        if not keep_segments:                      # no half-written segments left behind
            for c in children_cfg:
                seg = c["hip_raw_out"] if cfg.get("hip_raw_out") else c["out_path"]
                try:
                    os.remove(seg)
                except OSError:
                    pass
        raise RuntimeError("sharded export failed:\n" + "\n".join(failures))
    if cfg.get("hip_raw_out"):
        segs = [c["hip_raw_out"] for c in children_cfg]
        report = join_raw_segments(segs, str(cfg["hip_raw_out"]), keep_segments)
    else:
        # a shard with no frames (more GPUs than frames) starts no FFmpeg and leaves no segment
        segs = [c["out_path"] for c in children_cfg if os.path.exists(c["out_path"])]
        if not segs:
            raise RuntimeError("sharded export failed: no shard produced a segment (empty track?)")
        join_encoded_segments(segs, cfg["audio_path"], _with_ext(str(cfg["out_path"])), keep_segments)
        report = [(p, None, None) for p in segs]
    if progress_cb is not None:
        try:
            progress_cb(100)
        except Exception:
            pass
    return report


def export_tracks(cfgs, world: int, progress_cb=None, worker_cmd=None, devices=None, prestart_pct=None):
    """Many tracks: GPU g exports tracks g, g + world, ... one after another (``sharding.track_assignment``), the next
    track's worker started when the current one reports ``prestart_pct`` per cent (default 96, ``AUR_TRACK_PRESTART_PCT``;
    101 = strictly one after another)."""
    world = int(world)
    if prestart_pct is None:
        try:
            prestart_pct = int(os.environ.get("AUR_TRACK_PRESTART_PCT", "96"))
        except ValueError:
            prestart_pct = 96
    devices = list(devices) if devices else list(range(world))
    n = len(cfgs)
    finished = [0]
    failures, lock = [], threading.Lock()

    def gpu_lane(g, tmp):
        # One worker process per track (the reference's protocol: one config, one process).  A worker spends its first
        # seconds on the CPU -- interpreter, torch, HIP start-up, decoding and uploading the track -- so the next track's
        # worker is started while the current one renders its last few per cent: the GPU goes from one track's frames to
        # the next one's without the start-up in between (about 4 s per 94 s track of configs[4]).
        queue = sharding.track_assignment(n, world, g)
        running, nxt = [], 0

        def start(k):
            c = copy.deepcopy(cfgs[k])
            vs = dict(c.get("view_state") or {})
            vs.pop("hip_gpus", None)
            vs["hip_device"] = int(devices[g % len(devices)])
            c["view_state"] = vs
            return _Child(c, tmp, f"track{k:04d}", worker_cmd or default_worker_cmd(), lambda: None)

        def abort():
            for ch in running:
                ch.stop()
            for ch in running:
                ch.wait()

        while nxt < len(queue) or running:
            with lock:
                if failures:
                    abort()
                    return
            if nxt < len(queue) and (not running or (len(running) == 1 and running[0].pct >= prestart_pct)):
                running.append(start(queue[nxt]))
                nxt += 1
                continue
            ch = running[0]
            if ch.proc.poll() is None:
                time.sleep(0.05)
                continue
            msg = ch.failure(ch.wait())
            running.pop(0)
            with lock:
                if msg:
                    failures.append(msg)
                else:
                    finished[0] += 1
                    if progress_cb is not None:
                        try:
                            progress_cb(int(finished[0] * 100 / n))
                        except Exception:
                            pass
            if msg:
                abort()
                return

    with tempfile.TemporaryDirectory(prefix="aurora_tracks_") as tmp:
        lanes = [threading.Thread(target=gpu_lane, args=(g, tmp), daemon=True) for g in range(world)]
        for t in lanes:
            t.start()
        for t in lanes:
            t.join()
    if failures:
        raise RuntimeError("export of %d track(s) failed:\n%s" % (len(failures), "\n".join(failures)))
    return finished[0]


def main(argv) -> int:
    ap = argparse.ArgumentParser(prog="aurora_amd.export.sharded", description=__doc__.split("\n")[0])
    ap.add_argument("configs", nargs="+", help="worker config JSON file(s): one = time shards, several = whole tracks")
    ap.add_argument("--gpus", type=int, required=True)
    ap.add_argument("--devices", default="", help="comma-separated HIP device indices (default 0..gpus-1)")
    ap.add_argument("--keep-segments", action="store_true")
    try:
        args = ap.parse_args(argv[1:])
    except SystemExit:
        _emit({"type": "error", "message": "bad arguments"})
        return 2
    cfgs = []
    for p in args.configs:
        try:
            cfgs.append(json.loads(Path(p).expanduser().read_text(encoding="utf-8")))
        except Exception as e:
            _emit({"type": "error", "message": f"Invalid config {p}: {e}"})
            return 2
    devices = [int(v) for v in args.devices.split(",") if v.strip()] or None

    def progress(pct):
        _emit({"type": "progress", "pct": int(pct)})

    try:
        if len(cfgs) == 1:
            export_sharded(cfgs[0], args.gpus, progress, devices=devices, keep_segments=args.keep_segments)
        else:
            export_tracks(cfgs, args.gpus, progress, devices=devices)
    except Exception as e:
        _emit({"type": "error", "message": str(e)})
        return 1
    _emit({"type": "done"})
    return 0


if __name__ == "__main__":
    raise SystemExit(main(sys.argv))
