#!/usr/bin/env python3
"""Which memory can the product's pipe sink splice into a pipe?  aur_host_alloc (hipHostMalloc) memory and ordinary
pageable memory of several sizes: GB/s into a reading child, how much of it was spliced, errno on refusal, and the
device-to-host copy rate into the pinned kind.  (Anonymous memory registered with hipHostRegister: the pool refuses the
registration, tools/probes/hostreg_probe.hip.)"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (HIP runtime first)
from aurora_amd import _lib  # noqa: E402
from aurora_amd.export.exporter import _PipeSink, _grow_pipe  # noqa: E402
import numpy as np  # noqa: E402

lib = _lib.load()
torch.cuda.init()


def run(nbytes, reps, kind):
    p = C.c_void_p()
    if kind == "pinned":
        _lib.check(lib.aur_host_alloc(C.byref(p), nbytes), "aur_host_alloc")
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))
    else:
        arr = np.empty(nbytes, np.uint8)
    arr[:] = 7
    d2h = None
    if kind != "pageable":                       # device-to-host copy rate into this memory
        src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        ctx = _lib.Context(0)
        st = torch.cuda.Stream()
        for k in range(4):
            if k == 1:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.check(lib.aur_readback_async(ctx.handle, C.c_void_p(src.data_ptr()), nbytes, p, C.c_void_p(st.cuda_stream)), "rb")
        torch.cuda.synchronize()
        d2h = round(3 * nbytes / (time.perf_counter() - t0) / 1e9, 2)
        del src
    child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "pipe_drain.py")], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, bufsize=10 ** 7)
    _grow_pipe(child.stdin)
    sink = _PipeSink(child.stdin, splice=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        sink.write(memoryview(arr))
    child.stdin.close()
    rep = json.loads(child.stdout.read())
    child.wait()
    dt = time.perf_counter() - t0
    out = {"kind": kind, "bytes": nbytes, "reps": reps, "GBps": round(rep["bytes"] / dt / 1e9, 2),
           "spliced_frac": round(sink.spliced_bytes / rep["bytes"], 4), "refused_errno": getattr(sink, "refused_errno", None), "d2h_GBps": d2h}
    if kind == "pinned":
        lib.aur_host_free(p)

    print(json.dumps(out), flush=True)


for kind in ("pinned", "pageable"):
    for nbytes, reps in ((11 << 20, 200), (256 << 20, 16), (1592524800, 3)):
        run(nbytes, reps, kind)
