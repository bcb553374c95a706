"""Drop-in for the reference's ``audio/analysis.py`` on MI355X.

``Analyzer`` keeps the reference class's constructor, ``compute`` signature, return types and public
attributes (audio/analysis.py:5-33): ``sample_rate, fft_size, window, prev_mag, flux_smooth, flux_alpha,
hop, buffer``.  ``compute(x)`` slides ``buffer`` on the host exactly like the reference (both branches,
:18-23) and runs window -> rfft -> |X| -> flux -> log1p in one HIP launch (``aur_stft_buffer``).

The export path does not call ``compute`` per frame: ``compute_track`` analyses a whole track (or a time
shard of it) in one fused launch (``aur_stft_frames``), which is what ``export.exporter`` uses.

No CPU fallback: constructing an ``Analyzer`` without the HIP library or a GPU raises.
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from ..engine import get_engine


class Analyzer:
    def __init__(self, sample_rate=48000, fft_size=2048, device: int = 0):
        if int(fft_size) != _lib.FFT_SIZE:
            # the reference only ever constructs Analyzer(fft_size=2048) (export/exporter.py:114, audio/input.py:48)
            raise ValueError(f"fft_size {fft_size} unsupported on the HIP path (2048 only)")
        self.sample_rate = sample_rate
        self.fft_size = int(fft_size)
        self._eng = get_engine(device)
        torch = self._eng.torch
        self.window = self._eng.ctx.window()                       # float32[2048], periodic Hann
        self.flux_smooth = 0.0
        self.flux_alpha = 0.9
        self.hop = self.fft_size // 2
        self.buffer = np.zeros(self.fft_size, dtype=np.float32)
        self._d_buffer = self._eng.empty((self.fft_size,), torch.float32)
        self._d_prev_mag = self._eng.torch.zeros((_lib.NBINS,), dtype=torch.float32, device=self._eng.device)
        # log-spectrum and flux sum share one device buffer: one readback per compute()
        self._d_out = self._eng.empty((_lib.NBINS + 1,), torch.float32)
        self._d_logspec = self._d_out[:_lib.NBINS]
        self._d_flux = self._d_out[_lib.NBINS:]

    # prev_mag lives on the device; expose the reference attribute as a host copy
    @property
    def prev_mag(self) -> np.ndarray:
        return self._d_prev_mag.cpu().numpy()

    @prev_mag.setter
    def prev_mag(self, value):
        v = np.ascontiguousarray(value, dtype=np.float32)
        if v.shape != (_lib.NBINS,):
            raise ValueError("prev_mag must have shape (1025,)")
        self._d_prev_mag.copy_(self._eng.torch.from_numpy(v))

    def compute(self, x):
        n = len(x)
        x = np.asarray(x, dtype=np.float32)
        if n >= self.hop:                                          # analysis.py:18-20
            self.buffer[:-self.hop] = self.buffer[self.hop:]
            self.buffer[-self.hop:] = x[-self.hop:]
        else:                                                      # analysis.py:21-23
            self.buffer = np.roll(self.buffer, -n)
            self.buffer[-n:] = x
        torch = self._eng.torch
        self._d_buffer.copy_(torch.from_numpy(self.buffer))
        self._eng.stft_buffer(self._d_buffer, self._d_prev_mag, self._d_logspec, self._d_flux)
        out = self._d_out.cpu().numpy()
        spec = out[:_lib.NBINS].copy()                             # fresh float32[1025] per call
        flux = float(out[_lib.NBINS]) / (self.fft_size // 2 + 1)   # analysis.py:29
        self.flux_smooth = self.flux_alpha * self.flux_smooth + (1 - self.flux_alpha) * flux   # analysis.py:30
        return spec, self.flux_smooth

    def compute_track(self, audio, fps, frame_begin=0, n_frames=None, sensitivity=1.0):
        """Batched export analysis of a whole track (or frames [frame_begin, frame_begin + n_frames)).

        ``audio``: float32 [n] or [n, channels] (numpy, uploaded once) or a device tensor.
        Returns dict(pos, hop, logspec[F,1025], flux_smooth[F], energy[F], samples[F,1024]) as numpy arrays,
        with the values ``_read_frame`` + ``compute`` would produce frame by frame
        (export/exporter.py:227-238).
        """
        eng = self._eng
        track = (eng.upload_track(audio, self.sample_rate, fps) if isinstance(audio, np.ndarray)
                 else eng.make_track(audio, self.sample_rate, fps))
        params = _lib.ScanParams.defaults(sensitivity)
        params.flux_alpha = float(self.flux_alpha)
        state = eng.carried_state(track, frame_begin, params) if frame_begin > 0 else None
        an = eng.analyze(track, frame_begin, n_frames, want_samples=True)
        sc = eng.scan(an, params, state)
        return dict(pos=track.pos_host[an.frame_begin:an.frame_begin + an.n_frames], hop=track.hop,
                    logspec=an.logspec.cpu().numpy(), flux_smooth=sc.flux_smooth.cpu().numpy(),
                    energy=sc.energy.cpu().numpy(), samples=an.samples.cpu().numpy(),
                    pen_rgb=sc.pen_rgb.cpu().numpy()[:, :3], phase=sc.phase.cpu().numpy())
