"""aur_raster_spans_static (the entry point that allocates nothing) against aur_raster_spans_ws, through the C ABI.

Two calls in a row on one GPU, the second with a smaller input: the static work area keeps the first call's contents,
so the second call's frames show whether anything stale is read.  Bar: byte-identical (integer / byte work).
Collected last (file name) so a failure here cannot hide another test under `pytest -x`.

Why comparing two HIP entry points is enough here: `aur_raster_spans_ws` is itself pinned to the definition --
`test_linear_fill_equals_tile_fill` and the raster tests of tests/test_gpu_parity.py hold it bit-exact against the numpy
oracle (oracle/raster_ref.py) over sizes, span counts and random spans -- and the static entry point runs the very same
kernels on the very same arguments except for where the work area lives.  What this file has to show is that the place
makes no difference: stale contents, the size limit, and calls that would overlap on one GPU.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_spans(rng, F, K, W, H):
    sp = np.zeros((F, K, W), dtype=np.uint32)
    for f in range(F):
        for x in rng.choice(W, size=max(4, W // 5), replace=False):
            r0 = int(rng.integers(0, H))
            r1 = int(min(H, r0 + rng.integers(1, max(2, H // 2))))
            kind = int(rng.integers(0, 3))
            if kind == 0:                                      # a 2-px bar
                sp[f, 0, x] = r0 | (r1 << 16)
                if x > 0:
                    sp[f, 1, x - 1] = r0 | (r1 << 16)
            elif kind == 1 and r1 - r0 >= 3:                   # two disjoint intervals in one pixel column
                sp[f, 0, x] = r0 | ((r0 + 1) << 16)
                sp[f, K - 1, x] = (r1 - 1) | (r1 << 16)
            else:
                sp[f, int(rng.integers(0, K)), x] = r0 | (r1 << 16)
    return sp


def _call_pair(eng, sp, K, W, H, rng):
    import torch
    F = sp.shape[0]
    lib, h = eng.lib, eng.ctx.handle
    spans = torch.from_numpy(sp.view(np.int32)).to(eng.device)
    pen = torch.from_numpy(rng.integers(0, 256, size=(F, 4), dtype=np.uint8)).to(eng.device)
    pen[:, 3] = 0
    want = eng.fill(spans, K, pen, W, H)                       # aur_raster_spans_ws with a caller work area
    got = torch.full((F, H, W, 3), 0xA5, dtype=torch.uint8, device=eng.device)
    torch.cuda.synchronize()
    rc = lib.aur_raster_spans_static(h, C.c_void_p(spans.data_ptr()), K, C.c_void_p(pen.data_ptr()), F, W, H,
                                     C.c_void_p(got.data_ptr()), eng.stream())
    assert rc == 0, lib.aur_last_error()
    torch.cuda.synchronize()
    return got.cpu().numpy(), want.cpu().numpy()


@pytest.mark.parametrize("K", [2, 5])
def test_static_work_area_two_calls(K):
    from aurora_amd.engine import get_engine
    eng = get_engine(0)
    rng = np.random.default_rng(1000 + K)
    # first call: a large batch that nearly fills the static area; second: fewer, narrower frames (the stale summaries
    # of the first call sit in the same bytes with another layout); third: a width the linear fill does not take
    for (F, W, H) in ((140, 3840, 64), (3, 1280, 720), (2, 1000, 48)):
        sp = _random_spans(rng, F, K, W, H)
        got, want = _call_pair(eng, sp, K, W, H, rng)
        assert np.array_equal(got, want), (F, W, H, int(np.sum(got != want)))


def test_static_work_area_limit():
    from aurora_amd import _lib
    from aurora_amd.engine import get_engine
    eng = get_engine(0)
    lib, h = eng.lib, eng.ctx.handle
    need = C.c_size_t()
    assert lib.aur_raster_work_bytes(149, 3840, C.byref(need)) == 0 and need.value > _lib.STATIC_WORK_BYTES
    assert lib.aur_raster_work_bytes(148, 3840, C.byref(need)) == 0 and need.value <= _lib.STATIC_WORK_BYTES
    # rejected before anything is launched: the pointers are never dereferenced
    rc = lib.aur_raster_spans_static(h, C.c_void_p(4096), 2, C.c_void_p(4096), 149, 3840, 2160, C.c_void_p(4096), None)
    assert rc == -1 and b"static" in lib.aur_last_error() and b"148" in lib.aur_last_error()


def test_static_work_area_refuses_a_call_that_would_overlap_on_another_stream():
    """Every call on a GPU shares that GPU's work area.  A call on another stream while the previous one has not finished
    is refused (AUR_EINVAL naming both streams) instead of racing; once the first has finished -- or on the same stream,
    which orders the two -- the call goes through and its frames are right."""
    import torch
    from aurora_amd.engine import get_engine
    eng = get_engine(0)
    lib, h = eng.lib, eng.ctx.handle
    rng = np.random.default_rng(77)
    K, F, W, H = 2, 8, 1280, 360
    sp = _random_spans(rng, F, K, W, H)
    spans = torch.from_numpy(sp.view(np.int32)).to(eng.device)
    pen = torch.from_numpy(rng.integers(0, 256, size=(F, 4), dtype=np.uint8)).to(eng.device)
    pen[:, 3] = 0
    want = eng.fill(spans, K, pen, W, H)
    a, b = torch.cuda.Stream(device=eng.device), torch.cuda.Stream(device=eng.device)
    got_a = torch.zeros((F, H, W, 3), dtype=torch.uint8, device=eng.device)
    got_b = torch.zeros_like(got_a)
    torch.cuda.synchronize()

    def call(stream, out):
        return lib.aur_raster_spans_static(h, C.c_void_p(spans.data_ptr()), K, C.c_void_p(pen.data_ptr()), F, W, H,
                                           C.c_void_p(out.data_ptr()), C.c_void_p(stream.cuda_stream))

    with torch.cuda.stream(a):
        torch.cuda._sleep(400_000_000)                 # keeps stream a busy for a fraction of a second
    assert call(a, got_a) == 0, lib.aur_last_error()   # queued behind the sleep: not finished when the next call arrives
    rc = call(b, got_b)
    msg = lib.aur_last_error().decode()
    assert rc == -1 and "another stream" in msg, (rc, msg)
    assert f"{a.cuda_stream:x}" in msg.lower() and f"{b.cuda_stream:x}" in msg.lower(), msg
    assert call(a, got_a) == 0                         # the same stream is always fine: the stream orders the calls
    torch.cuda.synchronize()
    assert call(b, got_b) == 0, lib.aur_last_error()   # the previous call has finished: another stream may use the area
    torch.cuda.synchronize()
    assert torch.equal(got_a, want) and torch.equal(got_b, want)
