#!/usr/bin/env python3
"""How fast can a process hand frames to a child over a pipe?  write() vs vmsplice(), pipe sizes, write sizes.
CPU-only experiment behind the pipe sink of aurora_amd/export/exporter.py; results under profiles/."""
import ctypes as C
import fcntl
import json
import mmap
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
libc = C.CDLL(None, use_errno=True)


class iovec(C.Structure):
    _fields_ = [("iov_base", C.c_void_p), ("iov_len", C.c_size_t)]


def run(total, slot_bytes, pipe_size, how, chunk):
    slot = mmap.mmap(-1, slot_bytes)
    slot.write(b"\x5a" * slot_bytes)
    addr = C.addressof(C.c_char.from_buffer(slot))
    child = subprocess.Popen([sys.executable, os.path.join(HERE, "pipe_drain.py")], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, bufsize=0)
    fd = child.stdin.fileno()
    if pipe_size:
        fcntl.fcntl(fd, 1031, pipe_size)
    got = fcntl.fcntl(fd, 1032)
    mv = memoryview(slot)
    t0 = time.perf_counter()
    sent = 0
    while sent < total:
        off = 0
        while off < slot_bytes:
            n = min(chunk or slot_bytes, slot_bytes - off)
            if how == "write":
                k = os.write(fd, mv[off:off + n])
            else:
                iov = iovec(addr + off, n)
                k = libc.vmsplice(fd, C.byref(iov), 1, 0)
                if k < 0:
                    raise OSError(C.get_errno(), os.strerror(C.get_errno()))
            off += k
        sent += slot_bytes
    child.stdin.close()
    rep = json.loads(child.stdout.read())
    child.wait()
    dt = time.perf_counter() - t0
    assert rep["bytes"] == sent
    return {"how": how, "pipe": got, "chunk": chunk, "GBps": round(sent / dt / 1e9, 2)}


if __name__ == "__main__":
    total = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 8 * 10**9
    slot = 256 << 20
    for how in ("write", "vmsplice"):
        for pipe_size in (0, 1 << 20):
            for chunk in (0, 1 << 20, 1 << 16):
                print(json.dumps(run(total, slot, pipe_size, how, chunk)), flush=True)
