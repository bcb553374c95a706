"""The boundary the reference ends at: `proc.stdin.write` into FFmpeg (export/exporter.py:532-542).  The product's pipe
sink (`exporter._PipeSink` behind `_grow_pipe`) feeds a child that reads stdin as FFmpeg does and reports a sha256; the
digest must be that of the same export into an in-memory sink -- with whole ring slots spliced into the pipe from the
pinned ring (slots reused while the reader is still behind) and with the plain `write` path."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _DigestSink:
    def __init__(self):
        self.h, self.bytes = hashlib.sha256(), 0

    def write(self, buf):
        self.h.update(buf)
        self.bytes += len(buf)


@pytest.fixture(scope="module")
def eng():
    from aurora_amd.engine import get_engine
    return get_engine(0)


def _piped_digest(pipe, splice, chunk=None):
    from aurora_amd.export.exporter import _PipeSink, _grow_pipe
    cmd = [sys.executable, os.path.join(REPO, "tools", "pipe_drain.py"), "--sha256"]
    if chunk:
        cmd += ["--chunk", str(chunk)]
    child = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=10 ** 7)
    cap = _grow_pipe(child.stdin)
    sink = _PipeSink(child.stdin, splice=splice)
    n = pipe.run(sink)
    child.stdin.close()
    rep = json.loads(child.stdout.read())
    sink.close()
    assert child.wait() == 0
    return n, rep, sink, cap


@pytest.mark.parametrize("mode,wire", [("Spectrum - Linear", "rgb24"), ("Waveform - Linear", "yuv420p"),
                                       ("Spectrum - Radial", "rgb24")])
def test_piped_stream_equals_the_null_sink_stream(eng, mode, wire):
    from aurora_amd import synth
    from aurora_amd.pipeline import FramePipeline
    sr, fps, W, H = 48000, 60, 1280, 720
    audio = synth.pink(sr * 4 + 321, 2, seed=17)
    track = eng.upload_track(audio, sr, fps)
    # 5.5 - 11 MB batches in 3 slots: every slot is at least 4 pipe capacities, and is reused dozens of times per export
    make = lambda: FramePipeline(eng, track, mode, W, H, batch_frames=4, slots=3, wire_format=wire)
    ref = _DigestSink()
    n_ref = make().run(ref)
    assert n_ref == track.total_frames
    for splice, chunk in ((True, None), (True, 4096), (False, None)):      # a slow 4 KiB reader keeps the pipe full
        n, rep, sink, cap = _piped_digest(make(), splice, chunk)
        assert n == n_ref and rep["bytes"] == ref.bytes
        assert rep["sha256"] == ref.h.hexdigest(), f"{mode} {wire} splice={splice} chunk={chunk}: piped bytes differ"
        if splice and sink._fd is not None:
            # the pinned ring is a device-file mapping (vmsplice: EFAULT), so the sink stages it; either way no plain write
            assert sink.spliced_bytes > 0.8 * ref.bytes
            assert sink._staged == (sink.refused_errno is not None)
        if not splice:
            assert sink.spliced_bytes == 0


def test_exporter_run_pipe_end_to_end_through_a_reading_child(eng, tmp_path):
    """Exporter._run_pipe (exporter.py:531-567) on the HIP renderer with a stand-in FFmpeg that stores its stdin: the file
    holds the frames of the in-memory export."""
    from aurora_amd import synth
    from aurora_amd.export.exporter import Exporter
    sr, fps, W, H = 44100, 30, 1280, 720
    audio = synth.sweep(sr * 2, sr, 2)
    ex = Exporter((audio, sr), width=W, height=H, fps=fps, mode="Spectrum - Linear", view_state={"hip_batch_frames": 5})
    ref = _DigestSink()
    ex.render_to_sink(ref)
    out = tmp_path / "frames.bin"
    keep = [sys.executable, "-c", f"import sys, shutil; shutil.copyfileobj(sys.stdin.buffer, open({str(out)!r}, 'wb'), 1 << 20)"]
    seen = []
    ex._run_pipe(ex.make_renderer(), keep, progress_cb=seen.append)
    assert seen[-1] == 100 and seen == sorted(seen)
    assert hashlib.sha256(out.read_bytes()).hexdigest() == ref.h.hexdigest()
    assert out.stat().st_size == 60 * W * H * 3
