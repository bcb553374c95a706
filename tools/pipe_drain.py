#!/usr/bin/env python3
"""Stand-in for FFmpeg at the far end of the export pipe (export/exporter.py:532-542): reads stdin to its end the way
FFmpeg's rawvideo demuxer does -- blocking read() calls into its own buffer, one copy out of the pipe -- and prints one
JSON line {"bytes", "seconds", "sha256"?}.  Measurement / test tool: nothing in the product imports it.

    pipe_drain.py [--sha256] [--chunk BYTES]
"""
import hashlib
import json
import sys
import time


def main(argv):
    want_digest = "--sha256" in argv
    chunk = 1 << 22
    if "--chunk" in argv:
        chunk = int(argv[argv.index("--chunk") + 1])
    buf = bytearray(chunk)
    view = memoryview(buf)
    read = sys.stdin.buffer.raw.readinto
    h = hashlib.sha256() if want_digest else None
    n, t0 = 0, None
    while True:
        k = read(buf)
        if not k:
            break
        if t0 is None:
            t0 = time.perf_counter()
        n += k
        if h is not None:
            h.update(view[:k])
    dt = time.perf_counter() - t0 if t0 is not None else 0.0
    out = {"bytes": n, "seconds": dt}
    if h is not None:
        out["sha256"] = h.hexdigest()
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:]))
