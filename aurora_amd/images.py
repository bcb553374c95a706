"""Host side of the export's two image inputs: the background image (`view_state["background_path"]`,
widgets/rt_widget.py:388-404, :505-575) and the centre image of "Spectrum - Radial"
(`view_state["center_image_path"]`, rt_widget.py:226-240, :1060-1090).

Only ingest and integer layout decisions live here (decode a file once per export, where Qt would put the scaled image);
scaling, compositing and blending run in the HIP kernels of csrc/image.hip.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from dataclasses import dataclass

import numpy as np


def premultiply_argb(rgba: np.ndarray) -> np.ndarray:
    """uint8 [h, w, 4] straight (r, g, b, a) -> uint32 [h, w] premultiplied 0xAARRGGBB (Qt's qPremultiply: what
    QImage::smoothScaled converts an ARGB32 image to before it scales it; an opaque image is unchanged)."""
    p = rgba.astype(np.uint32)
    a = p[..., 3]
    out = a << 24
    for ch, shift in ((0, 16), (1, 8), (2, 0)):
        t = p[..., ch] * a
        out |= ((t + (t >> 8) + 0x80) >> 8) << shift
    return np.ascontiguousarray(out.astype(np.uint32))


def load_image_rgba(path: str) -> np.ndarray:
    """Decoded image as uint8 [h, w, 4] straight alpha (QImage(path), rt_widget.py:231, :404).  Pillow when it is
    installed, else FFmpeg (the reference already needs it for every export); 16-bit files are reduced to 8 bits."""
    path = os.path.abspath(os.path.expanduser(str(path)))
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    try:
        from PIL import Image
    except ImportError:
        Image = None
    if Image is not None:
        with Image.open(path) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))
    try:
        from .export.exporter import _ffmpeg_bin
        ffmpeg = _ffmpeg_bin()                     # AURORA_FFMPEG, IMAGEIO_FFMPEG_EXE, imageio_ffmpeg, PATH
    except Exception:
        ffmpeg = shutil.which("ffmpeg")
    if ffmpeg is None:
        raise RuntimeError("neither Pillow nor ffmpeg is available to decode " + path)
    # one decoded frame as raw rgba on stdout; the size is in FFmpeg's description of the input stream on stderr
    # ("Stream #0:0: Video: png, rgba(pc), 640x480, ...") and must agree with the number of bytes that arrived
    run = subprocess.run([ffmpeg, "-hide_banner", "-nostdin", "-i", path, "-f", "rawvideo", "-pix_fmt", "rgba",
                          "-frames:v", "1", "-"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if run.returncode != 0:
        raise RuntimeError(f"ffmpeg could not decode {path}: {run.stderr.decode('utf-8', 'ignore')[-2000:]}")
    import re
    for line in run.stderr.splitlines():
        if b"Video:" not in line:
            continue
        for m in re.finditer(rb"(?<![\dx])(\d{1,5})x(\d{1,5})(?![\dx])", line):
            w, h = int(m.group(1)), int(m.group(2))
            if w > 0 and h > 0 and w * h * 4 == len(run.stdout):
                return np.frombuffer(run.stdout, dtype=np.uint8).reshape(h, w, 4).copy()
    raise RuntimeError(f"ffmpeg did not report a size that matches the {len(run.stdout)} decoded bytes of {path}")


def qsize_scaled(w: int, h: int, tw: int, th: int, aspect: str):
    """QSize(w, h).scaled(QSize(tw, th), aspect) with Qt's integer arithmetic; aspect in {"ignore", "keep", "expand"},
    then QImage::scaled's clamp to at least 1x1."""
    if aspect == "ignore" or w == 0 or h == 0:
        nw, nh = tw, th
    else:
        rw = (th * w) // h
        use_height = (rw <= tw) if aspect == "keep" else (rw >= tw)
        nw, nh = (rw, th) if use_height else (tw, (tw * h) // w)
    return max(nw, 1), max(nh, 1)


@dataclass
class BackgroundConfig:
    """`BackgroundConfig` as the exporter fills it from view_state (export/exporter.py:141-151) and `set_background_config`
    stores it (rt_widget.py:388-404)."""
    path: str | None = None
    scale_mode: str = "fill"
    offset_x: int = 0
    offset_y: int = 0
    dim_percent: int = 0

    @classmethod
    def from_view_state(cls, vs):
        vs = vs or {}
        return cls(path=vs.get("background_path") or None,
                   scale_mode=vs.get("background_scale_mode", "fill"),
                   offset_x=int(vs.get("background_offset_x", 0) or 0),
                   offset_y=int(vs.get("background_offset_y", 0) or 0),
                   dim_percent=int(vs.get("background_dim_percent", 0) or 0))

    @property
    def active(self):
        return bool(self.path)

    def plan(self, img_w: int, img_h: int, W: int, H: int):
        """(scaled_w, scaled_h, tile, x, y, dim_alpha): `_draw_bg_image` / `_scaled_bg_for` (rt_widget.py:519-575)."""
        mode = (self.scale_mode or "fill").lower()
        d = max(0, min(100, int(self.dim_percent)))
        dim_alpha = int(255 * (d / 100.0)) if d > 0 else 0
        if mode == "tile":
            return img_w, img_h, True, self.offset_x, self.offset_y, dim_alpha
        if mode == "center":
            sw, sh = img_w, img_h
        else:
            aspect = {"fit": "keep", "stretch": "ignore"}.get(mode, "expand")
            sw, sh = qsize_scaled(img_w, img_h, W, H, aspect)
        x = (W - sw) // 2 + self.offset_x
        y = (H - sh) // 2 + self.offset_y
        return sw, sh, False, x, y, dim_alpha
