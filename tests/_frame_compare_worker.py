"""Worker side of tests/test_gpu_e2e_parity.py: renders one frame with the ORACLE fed the oracle's own analysis values and
compares it with the GPU frame parked in shared memory.  Runs in spawned CPU processes (numpy + oracle only, no torch)."""
import numpy as np


def compare_frame(task):
    """-> (pixels that differ, pixels that differ by more than 1 LSB in a channel, PSNR in dB)."""
    from multiprocessing import shared_memory
    from oracle import raster_ref as R
    shm_name, offset, mode, samples, spectrum, energy, pen, W, H = task
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        got = np.ndarray((H, W, 3), dtype=np.uint8, buffer=shm.buf, offset=offset)
        want = R.render_frame(mode, samples, spectrum, energy, pen, W, H)
        if np.array_equal(got, want):
            res = (0, 0, float("inf"))
        else:
            d = np.abs(got.astype(np.int16) - want.astype(np.int16)).max(axis=2)
            res = (int(np.count_nonzero(d)), int(np.count_nonzero(d > 1)), float(R.psnr(got, want)))
        del got
        return res
    finally:
        shm.close()


def compare_closed_frame(task):
    """The closed-path modes: the oracle's own vertices (from the oracle's own analysis) rasterised by the oracle, against
    the GPU frame in shared memory.  Same triple as compare_frame."""
    from multiprocessing import shared_memory
    from oracle import raster_ref as R
    shm_name, offset, mode, path, env, pen, W, H = task
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        got = np.ndarray((H, W, 3), dtype=np.uint8, buffer=shm.buf, offset=offset)
        if mode == "Spectrum - Radial":
            want = R.render_radial_frame(path, env, pen, W, H)
        else:
            want = R.render_wave_circular_frame(path, pen, W, H)
        if np.array_equal(got, want):
            res = (0, 0, float("inf"))
        else:
            d = np.abs(got.astype(np.int16) - want.astype(np.int16)).max(axis=2)
            res = (int(np.count_nonzero(d)), int(np.count_nonzero(d > 1)), float(R.psnr(got, want)))
        del got
        return res
    finally:
        shm.close()


def compare_union_frame(task):
    """BASELINE configs[3] union composite (particles, then the audio-reactive disc, then the six feather rings blended on
    top): the oracle's particle frame from `spectrum`, the disc polygon `env` filled opaque, the rings source-over in
    order.  `env` / `rings` are whatever the caller wants the composite pinned to (the GPU's own vertices: bit-exact
    expected; the oracle's own: a last-ulp vertex difference may flip an edge pixel).  Same triple as compare_frame."""
    from multiprocessing import shared_memory
    from oracle import raster_ref as R
    shm_name, offset, spectrum, energy, pen, env, rings, alphas, W, H = task
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        got = np.ndarray((H, W, 3), dtype=np.uint8, buffer=shm.buf, offset=offset)
        want = R.render_frame("Particles", np.zeros(1024, np.float32), spectrum, energy, pen, W, H)
        want[R.unpack_mask(R.polygon_mask(env, W, H), W)] = np.asarray(R.DISC_RGB, np.uint8)
        for verts, a in zip(rings, alphas):
            R.blend_solid(want, R.unpack_mask(R.polygon_mask(verts, W, H), W), R.BG, int(a))
        if np.array_equal(got, want):
            res = (0, 0, float("inf"))
        else:
            d = np.abs(got.astype(np.int16) - want.astype(np.int16)).max(axis=2)
            res = (int(np.count_nonzero(d)), int(np.count_nonzero(d > 1)), float(R.psnr(got, want)))
        del got
        return res
    finally:
        shm.close()
