// Antialiased centre disc, centre image and background image of the export (gfx950).
//
// Replaces, for a whole batch of frames per launch:
//   * widgets/rt_widget.py:1056-1096  `setRenderHint(Antialiasing)`, `setClipPath(path)`, then `fillPath(path, (30,30,40))`
//     or `drawImage(dx, dy, img.scaled(out_w, out_h, IgnoreAspectRatio, SmoothTransformation))`  -> disc_aa_kernel
//   * widgets/rt_widget.py:1065-1088  placement of the centre image (bounding rect of the clip path, zoom, motion)
//                                                                                                 -> image_placement_kernel
//   * widgets/rt_widget.py:505-575    `_draw_background` / `_draw_bg_image` / `_scaled_bg_for`      -> background_kernel
//   * QImage::scaled(..., Qt::SmoothTransformation) (Qt's qimagescale.cpp, the Imlib2 scaler)       -> scaled_pixel
//
// Qt's raster engine and image scaler are not part of the reference tree: the rules below are the ones oracle/raster_ref.py
// states (coverage_aa, smooth_scale_pixel, blend_cov, compose_background) and are "parity unpinned" against Qt; this file
// matches the oracle bit for bit.  Everything is integer arithmetic, so no result depends on the order of the atomics.
#include "common.h"

#include <math.h>

namespace {

// ---- antialiased even-odd coverage (oracle: coverage_aa) ---------------------------------------------------------------

__device__ __forceinline__ long long floordiv_pos(long long a, long long b) {      // floor(a / b), b > 0
    long long q = a / b;
    if ((a % b) != 0 && a < 0) --q;
    return q;
}

__device__ __forceinline__ bool fixed_26_6(double v, long long& out) {              // qRound(v * 64) in 1/256 px units
    if (!(v == v)) return false;
    double t = floor(v * 64.0 + 0.5);
    t = fmin(fmax(t, -1.0e9), 1.0e9);
    out = (long long)t * 4;
    return true;
}

// Accumulates the cells of one polygon edge that fall into pixel rows [r0, r0 + R) and columns [bx0, bx0 + bw):
// cover[row][0] collects everything left of the box, cover[row][1 + c] / area[row][c] the cells inside it.
__device__ void accumulate_edge(long long x0, long long y0, long long x1, long long y1, int r0, int R, int bx0, int bw,
                                int* cover, int* area) {
    if (y0 == y1) return;
    int sign = 1;
    if (y1 < y0) {
        long long t = x0; x0 = x1; x1 = t;
        t = y0; y0 = y1; y1 = t;
        sign = -1;
    }
    const long long dx = x1 - x0, dy = y1 - y0;
    long long xl, yl_x, xr, yr_x;                       // end points ordered by x, for the column crossings
    if (x0 <= x1) { xl = x0; yl_x = y0; xr = x1; yr_x = y1; }
    else          { xl = x1; yl_x = y1; xr = x0; yr_x = y0; }
    long long ey_lo = y0 >> 8, ey_hi = (y1 - 1) >> 8;
    if (ey_lo < r0) ey_lo = r0;
    if (ey_hi > r0 + R - 1) ey_hi = r0 + R - 1;
    for (long long ey = ey_lo; ey <= ey_hi; ++ey) {
        const long long ya = y0 > (ey << 8) ? y0 : (ey << 8);
        const long long yb = y1 < ((ey + 1) << 8) ? y1 : ((ey + 1) << 8);
        const long long xa = (ya == y0) ? x0 : x0 + floordiv_pos((ya - y0) * dx, dy);
        const long long xb = (yb == y1) ? x1 : x0 + floordiv_pos((yb - y0) * dx, dy);
        int* crow = cover + (int)(ey - r0) * (bw + 1);
        int* arow = area + (int)(ey - r0) * bw;
        if (xa == xb) {
            const long long ex = xa >> 8;
            const long long c = ex - bx0;
            if (c >= bw) continue;
            const int dc = sign * (int)(yb - ya);
            if (c < 0) {
                atomicAdd(crow, dc);
            } else {
                atomicAdd(crow + 1 + c, dc);
                atomicAdd(arow + c, dc * 2 * (int)(xa - (ex << 8)));
            }
            continue;
        }
        const bool right = xa < xb;
        const long long lo = right ? xa : xb, hi = right ? xb : xa;
        const long long ylo = right ? ya : yb, yhi = right ? yb : ya;      // y at the piece's left / right end
        long long ex_hi = (hi - 1) >> 8;
        if (ex_hi > (long long)bx0 + bw - 1) ex_hi = (long long)bx0 + bw - 1;   // cells right of the box change nothing
        long long ex_lo = lo >> 8;
        if (ex_lo < bx0) {
            // the cells left of the box all add to cover[row][0], and their shares telescope (a cell's right crossing is
            // the next cell's left crossing, same formula): one step to the box's left edge, however far out the edge starts
            const long long edge = (long long)bx0 << 8;
            const long long xcut = hi < edge ? hi : edge;
            long long pyb;
            if (xcut == hi) pyb = yhi;
            else {
                pyb = yl_x + floordiv_pos((xcut - xl) * (yr_x - yl_x), xr - xl);
                pyb = pyb < ya ? ya : (pyb > yb ? yb : pyb);
            }
            atomicAdd(crow, sign * (int)(right ? (pyb - ylo) : (ylo - pyb)));
            if (xcut == hi) continue;
            ex_lo = bx0;
        }
        for (long long ex = ex_lo; ex <= ex_hi; ++ex) {
            const long long cxa = lo > (ex << 8) ? lo : (ex << 8);
            const long long cxb = hi < ((ex + 1) << 8) ? hi : ((ex + 1) << 8);
            long long pya, pyb;
            if (cxa == lo) pya = ylo;
            else {
                pya = yl_x + floordiv_pos((cxa - xl) * (yr_x - yl_x), xr - xl);
                pya = pya < ya ? ya : (pya > yb ? yb : pya);
            }
            if (cxb == hi) pyb = yhi;
            else {
                pyb = yl_x + floordiv_pos((cxb - xl) * (yr_x - yl_x), xr - xl);
                pyb = pyb < ya ? ya : (pyb > yb ? yb : pyb);
            }
            const int d = sign * (int)(right ? (pyb - pya) : (pya - pyb));
            const long long c = ex - bx0;
            if (c < 0) {
                atomicAdd(crow, d);
            } else {
                atomicAdd(crow + 1 + c, d);
                atomicAdd(arow + c, d * (int)((cxa - (ex << 8)) + (cxb - (ex << 8))));
            }
        }
    }
}

// ---- QImage::scaled(SmoothTransformation) of a premultiplied ARGB32 image, one pixel at a time (oracle: smooth_scale_pixel)

struct Px {
    long long r, g, b, a;
    __device__ Px operator*(long long k) const { return {r * k, g * k, b * k, a * k}; }
    __device__ Px operator+(const Px& o) const { return {r + o.r, g + o.g, b + o.b, a + o.a}; }
    __device__ Px operator>>(int k) const { return {r >> k, g >> k, b >> k, a >> k}; }
};

__device__ __forceinline__ Px unpack_argb(uint32_t v) {
    return {(long long)((v >> 16) & 255u), (long long)((v >> 8) & 255u), (long long)(v & 255u), (long long)(v >> 24)};
}
__device__ __forceinline__ uint32_t pack_argb(const Px& p) {
    return (((uint32_t)p.a & 255u) << 24) | (((uint32_t)p.r & 255u) << 16) | (((uint32_t)p.g & 255u) << 8) |
           ((uint32_t)p.b & 255u);
}

struct Axis {
    int s, d, up, cp;
    long long inc, val0;
};

__device__ __forceinline__ Axis make_axis(int s, int d) {
    Axis a;
    a.s = s;
    a.d = d;
    a.up = d >= s;
    a.inc = ((long long)s << 16) / d;
    a.val0 = a.up ? (0x8000LL * s) / d - 0x8000LL : 0LL;
    a.cp = a.up ? 0 : (int)((((long long)d << 14) + s - 1) / s);
    return a;
}

__device__ __forceinline__ void axis_point(const Axis& a, int i, int& pos, int& ap) {
    const long long val = a.val0 + (long long)i * a.inc;
    if (a.up) {
        const long long p = val >> 16;
        pos = p < 0 ? 0 : (int)p;
        ap = (p < 0 || p >= a.s - 1) ? 0 : (int)((val >> 8) & 0xff);
    } else {
        pos = (int)(val >> 16);
        ap = (int)(((0x10000LL - (val & 0xffff)) * a.cp) >> 16) | (a.cp << 16);
    }
}

__device__ __forceinline__ Px interp256(const Px& x, long long a, const Px& y, long long b) {
    return (x * a + y * b) >> 8;
}

template <typename F>
__device__ __forceinline__ Px area_axis(F fetch, int start, int apw, int limit) {
    const int xyap = apw & 0xffff, c = apw >> 16;
    int i = start;
    Px acc = fetch(i < limit ? i : limit) * (long long)xyap;
    int j = (1 << 14) - xyap;
    while (j > c && c > 0) {
        ++i;
        acc = acc + fetch(i < limit ? i : limit) * (long long)c;
        j -= c;
    }
    ++i;
    acc = acc + fetch(i < limit ? i : limit) * (long long)j;
    return acc;
}

__device__ uint32_t scaled_pixel(const uint32_t* __restrict__ img, int sw, int sh, int dw, int dh, int x, int y) {
    const Axis ax = make_axis(sw, dw), ay = make_axis(sh, dh);
    int sx, xap, sy, yap;
    axis_point(ax, x, sx, xap);
    axis_point(ay, y, sy, yap);
    auto px = [&](int xx, int yy) { return unpack_argb(img[(long)yy * sw + xx]); };
    if (ax.up && ay.up) {
        if (yap > 0) {
            if (xap > 0) {
                const Px top = interp256(px(sx, sy), 256 - xap, px(sx + 1, sy), xap);
                const Px bot = interp256(px(sx, sy + 1), 256 - xap, px(sx + 1, sy + 1), xap);
                return pack_argb(interp256(top, 256 - yap, bot, yap));
            }
            return pack_argb(interp256(px(sx, sy), 256 - yap, px(sx, sy + 1), yap));
        }
        if (xap > 0) return pack_argb(interp256(px(sx, sy), 256 - xap, px(sx + 1, sy), xap));
        return img[(long)sy * sw + sx];
    }
    if (ax.up && !ay.up) {
        auto col = [&](int xx) { return area_axis([&](int i) { return px(xx, i); }, sy, yap, sh - 1); };
        Px r = col(sx);
        if (xap > 0) {
            const Px rr = col(sx + 1);
            r = (r * (long long)(256 - xap) + rr * (long long)xap) >> 8;
        }
        return pack_argb(r >> 14);
    }
    if (!ax.up && ay.up) {
        auto row = [&](int yy) { return area_axis([&](int i) { return px(i, yy); }, sx, xap, sw - 1); };
        Px r = row(sy);
        if (yap > 0) {
            const Px rr = row(sy + 1);
            r = (r * (long long)(256 - yap) + rr * (long long)yap) >> 8;
        }
        return pack_argb(r >> 14);
    }
    auto rowsum = [&](int yy) { return area_axis([&](int i) { return px(i, yy); }, sx, xap, sw - 1) >> 4; };
    return pack_argb(area_axis(rowsum, sy, yap, sh - 1) >> 24);
}

__global__ __launch_bounds__(256) void image_scale_kernel(const uint32_t* __restrict__ src, int sw, int sh,
                                                          uint32_t* __restrict__ dst, int dw, int dh) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    dst[(long)y * dw + x] = scaled_pixel(src, sw, sh, dw, dh, x, y);
}

// ---- centre image placement (rt_widget.py:1065-1088), one wave per frame -----------------------------------------------
__global__ __launch_bounds__(64) void image_placement_kernel(const double* __restrict__ verts, int nv, int verts_static,
                                                             const double* __restrict__ energy,
                                                             const double* __restrict__ phase, int img_w, int img_h,
                                                             double user_zoom, double cm, int* __restrict__ out) {
    const int f = blockIdx.x, lane = threadIdx.x;
    const double* V = verts + (verts_static ? 0L : (long)f * nv * 2);
    double minx = 1.0e300, maxx = -1.0e300, miny = 1.0e300, maxy = -1.0e300;
    for (int i = lane; i < nv; i += 64) {
        const double x = V[2 * i], y = V[2 * i + 1];
        minx = fmin(minx, x); maxx = fmax(maxx, x);
        miny = fmin(miny, y); maxy = fmax(maxy, y);
    }
    for (int m = 1; m < 64; m <<= 1) {
        minx = fmin(minx, __shfl_xor(minx, m, 64)); maxx = fmax(maxx, __shfl_xor(maxx, m, 64));
        miny = fmin(miny, __shfl_xor(miny, m, 64)); maxy = fmax(maxy, __shfl_xor(maxy, m, 64));
    }
    if (lane != 0) return;
    // path.boundingRect().adjusted(-2, -2, 2, 2)
    const double rx = minx + -2.0, ry = miny + -2.0;
    const double rw = ((maxx - minx) + 2.0) - -2.0, rh = ((maxy - miny) + 2.0) - -2.0;
    int rect_w = (int)rw, rect_h = (int)rh;
    if (rect_w < 1) rect_w = 1;
    if (rect_h < 1) rect_h = 1;
    double scale = fmax((double)rect_w / (double)img_w, (double)rect_h / (double)img_h);
    scale = scale * user_zoom;
    if (cm > 0.0) {
        const double motion_zoom = 1.0 + ((0.12 * cm) * (0.4 + 0.6 * energy[f])) * (0.5 + 0.5 * sin(phase[f] * 1.7));
        scale = scale * fmax(0.1, motion_zoom);
    }
    double ow = rint((double)img_w * scale), oh = rint((double)img_h * scale);     // round(): half to even
    if (ow < 1.0) ow = 1.0;
    if (oh < 1.0) oh = 1.0;
    ow = fmin(ow, 1.0e9);
    oh = fmin(oh, 1.0e9);
    const int out_w = (int)ow, out_h = (int)oh;
    const double cx_rect = rx + rw / 2.0, cy_rect = ry + rh / 2.0;
    int dx = (int)rint(cx_rect - (double)out_w / 2.0), dy = (int)rint(cy_rect - (double)out_h / 2.0);
    if (cm > 0.0) {
        const int j = (int)(cm * 6.0);
        dx += (int)rint((double)j * sin(phase[f] * 2.3));
        dy += (int)rint((double)j * cos(phase[f] * 1.9));
    }
    out[4 * f + 0] = out_w;
    out[4 * f + 1] = out_h;
    out[4 * f + 2] = dx;
    out[4 * f + 3] = dy;
}

// ---- the disc itself: coverage of R pixel rows in LDS, then the composite ------------------------------------------------

__device__ __forceinline__ uint32_t byte_mul(uint32_t x, uint32_t a) {
    const uint32_t t = x * a;
    return (t + (t >> 8) + 0x80u) >> 8;
}

struct DiscArgs {
    const double* verts;       // [F or 1][nv][2]
    int nv, verts_static;
    int W, H, W32;
    int bx0, by0, bw, bh;      // box the polygon lies in (clamped to the frame)
    int R;                     // pixel rows per workgroup
    int mode;                  // 0: colour, fill under its own clip (coverage squared); 1: image scaled per frame; 2: image already scaled
    uint32_t colour;           // premultiplied 0xAARRGGBB
    const uint32_t* image;     // premultiplied ARGB32 [img_h][img_w]
    int img_w, img_h;
    const int* placement;      // [F or 1][4] = out_w, out_h, dx, dy
    int placement_static;
    int skip_full;             // colour mode: pixels of full coverage were painted by the plane fill already
    uint32_t* full_plane;      // optional uint32 [F][H][W32]: bit = full coverage; words outside the box are not written
    uint8_t* frames;           // optional uint8 [F][H][W][3]
};

__global__ __launch_bounds__(256) void disc_aa_kernel(DiscArgs A) {
    extern __shared__ int s_acc[];
    const int f = blockIdx.y;
    const int r0 = A.by0 + blockIdx.x * A.R;                       // first pixel row of this workgroup
    const int rows = (A.by0 + A.bh - r0 < A.R) ? A.by0 + A.bh - r0 : A.R;
    int* cover = s_acc;                                            // [R][bw + 1]
    int* area = s_acc + A.R * (A.bw + 1);                          // [R][bw], later the coverage itself
    const int n_acc = A.R * (2 * A.bw + 1);
    for (int i = threadIdx.x; i < n_acc; i += 256) s_acc[i] = 0;
    __syncthreads();
    const double* V = A.verts + (A.verts_static ? 0L : (long)f * A.nv * 2);
    for (int e = threadIdx.x; e < A.nv; e += 256) {
        const int e1 = (e + 1 == A.nv) ? 0 : e + 1;
        long long x0, y0, x1, y1;
        if (!fixed_26_6(V[2 * e], x0) || !fixed_26_6(V[2 * e + 1], y0) || !fixed_26_6(V[2 * e1], x1) ||
            !fixed_26_6(V[2 * e1 + 1], y1))
            continue;
        accumulate_edge(x0, y0, x1, y1, r0, rows, A.bx0, A.bw, cover, area);
    }
    __syncthreads();
    // running sum of the cover along every row (one wave per row), folded into the 0..255 coverage
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = wave; row < rows; row += 4) {
        int carry = cover[row * (A.bw + 1)];
        for (int c0 = 0; c0 < A.bw; c0 += 64) {
            const int c = c0 + lane;
            int v = (c < A.bw) ? cover[row * (A.bw + 1) + 1 + c] : 0;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o, 64);
                if (lane >= o) v += t;
            }
            const int acc = carry + v;
            carry += __shfl(v, 63, 64);
            if (c < A.bw) {
                long long val = (long long)acc * 512 - area[row * A.bw + c];
                if (val < 0) val = -val;
                int cv = (int)(val >> 9) & 511;
                if (cv > 256) cv = 512 - cv;
                if (cv > 255) cv = 255;
                area[row * A.bw + c] = cv;
            }
        }
    }
    __syncthreads();
    if (A.full_plane != nullptr) {
        const int w0 = A.bx0 >> 5, w1 = (A.bx0 + A.bw - 1) >> 5;
        const int nw = w1 - w0 + 1;
        for (int i = threadIdx.x; i < rows * nw; i += 256) {
            const int row = i / nw, w = w0 + i % nw;
            uint32_t bits = 0u;
            for (int b = 0; b < 32; ++b) {
                const int c = 32 * w + b - A.bx0;
                if (c >= 0 && c < A.bw && area[row * A.bw + c] == 255) bits |= 1u << b;
            }
            A.full_plane[((long)f * A.H + (r0 + row)) * A.W32 + w] = bits;
        }
    }
    if (A.frames == nullptr) return;
    int out_w = 0, out_h = 0, pdx = 0, pdy = 0;
    if (A.mode != 0) {
        const int* p = A.placement + (A.placement_static ? 0 : 4 * f);
        out_w = p[0]; out_h = p[1]; pdx = p[2]; pdy = p[3];
    }
    for (int i = threadIdx.x; i < rows * A.bw; i += 256) {
        const int row = i / A.bw, c = i - row * A.bw;
        const uint32_t cv = (uint32_t)area[i];
        if (cv == 0u) continue;
        const int x = A.bx0 + c, y = r0 + row;
        uint32_t src, ce;
        if (A.mode == 0) {
            ce = byte_mul(cv, cv);                                 // fill coverage x clip coverage (qt_intersect_spans)
            if (ce == 0u || (A.skip_full && ce == 255u)) continue;
            src = A.colour;
        } else {
            const int ix = x - pdx, iy = y - pdy;
            if (ix < 0 || iy < 0 || ix >= out_w || iy >= out_h) continue;
            ce = cv;
            src = (A.mode == 1) ? scaled_pixel(A.image, A.img_w, A.img_h, out_w, out_h, ix, iy)
                                : A.image[(long)iy * out_w + ix];
        }
        const uint32_t sa = byte_mul(src >> 24, ce), sr = byte_mul((src >> 16) & 255u, ce);
        const uint32_t sg = byte_mul((src >> 8) & 255u, ce), sb = byte_mul(src & 255u, ce);
        uint8_t* px = A.frames + (((long)f * A.H + y) * A.W + x) * 3;
        if (sa == 255u) {
            px[0] = (uint8_t)sr; px[1] = (uint8_t)sg; px[2] = (uint8_t)sb;
        } else if (sa != 0u || (sr | sg | sb) != 0u) {
            const uint32_t inv = 255u - sa;
            px[0] = (uint8_t)(sr + byte_mul(px[0], inv));
            px[1] = (uint8_t)(sg + byte_mul(px[1], inv));
            px[2] = (uint8_t)(sb + byte_mul(px[2], inv));
        }
    }
}

// The same composite for a frame-independent disc: its partially covered pixels are a short list (offset in the frame,
// effective coverage), computed once per frame size; one lane per (frame, list entry).
__global__ __launch_bounds__(256) void blend_pixels_kernel(const uint32_t* __restrict__ offsets,
                                                           const uint8_t* __restrict__ cov, int n, uint32_t ca,
                                                           uint32_t cr, uint32_t cg, uint32_t cb, long frame_bytes,
                                                           uint8_t* __restrict__ frames) {
    // the colour arrives as four separate words: with the channels extracted from one packed word in here, hipcc
    // (ROCm 7.2) matched the four byte products to v_dot4_u32_u8 and produced wrong colours (as in aur_blend_planes)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t ce = cov[i];
    const uint32_t sa = byte_mul(ca, ce), sr = byte_mul(cr, ce);
    const uint32_t sg = byte_mul(cg, ce), sb = byte_mul(cb, ce);
    uint8_t* px = frames + (long)blockIdx.y * frame_bytes + (long)offsets[i] * 3;
    const uint32_t inv = 255u - sa;
    px[0] = (uint8_t)(sr + byte_mul(px[0], inv));
    px[1] = (uint8_t)(sg + byte_mul(px[1], inv));
    px[2] = (uint8_t)(sb + byte_mul(px[2], inv));
}

// ---- background image -> the frame every export frame starts from (rt_widget.py:829-830, :519-545) ---------------------
// base = fillRect (13,13,18); the image drawn at (x, y) (premultiplied source-over: PNGs with alpha keep the fill under
// them), or tiled from (off_x, off_y) at multiples of its size; then fillRect(0,0,w,h, QColor(0,0,0,dim_alpha)).
__global__ __launch_bounds__(256) void background_kernel(const uint32_t* __restrict__ img, int iw, int ih, int tile,
                                                         int ox, int oy, int dim_alpha, int W, int H,
                                                         uint8_t* __restrict__ base) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    uint32_t r = 13u, g = 13u, b = 18u;
    int ix = x - ox, iy = y - oy;
    bool hit;
    if (tile) {
        // tiles are drawn at (tx + off_x, ty + off_y) for tx = 0, sw, 2 sw, .. < w and ty likewise (:523-528): nothing left
        // of / above the first tile, and the last tile column / row may reach past the frame's own tiling range
        hit = ix >= 0 && iy >= 0 && (ix / iw) * iw < W && (iy / ih) * ih < H;
        ix %= iw;
        iy %= ih;
    } else {
        hit = ix >= 0 && iy >= 0 && ix < iw && iy < ih;
    }
    if (hit) {
        const uint32_t s = img[(long)iy * iw + ix];
        const uint32_t inv = 255u - (s >> 24);
        r = ((s >> 16) & 255u) + byte_mul(r, inv);
        g = ((s >> 8) & 255u) + byte_mul(g, inv);
        b = (s & 255u) + byte_mul(b, inv);
    }
    if (dim_alpha > 0) {                                           // premultiplied black: d' = BYTE_MUL(d, 255 - a)
        const uint32_t inv = 255u - (uint32_t)dim_alpha;
        r = byte_mul(r, inv);
        g = byte_mul(g, inv);
        b = byte_mul(b, inv);
    }
    uint8_t* px = base + ((long)y * W + x) * 3;
    px[0] = (uint8_t)r; px[1] = (uint8_t)g; px[2] = (uint8_t)b;
}

}  // namespace

extern "C" int aur_image_scale(aur_ctx* ctx, const uint32_t* d_src, int src_w, int src_h, uint32_t* d_dst, int dst_w,
                               int dst_h, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(src_w >= 1 && src_h >= 1 && src_w <= 32767 && src_h <= 32767, "aur_image_scale: source %dx%d outside [1..32767]^2",
                src_w, src_h);
    AUR_REQUIRE(dst_w >= 1 && dst_h >= 1 && dst_w <= 32767 && dst_h <= 32767, "aur_image_scale: target %dx%d outside [1..32767]^2",
                dst_w, dst_h);
    AUR_REQUIRE(d_src && d_dst, "aur_image_scale: null device pointer");
    hipLaunchKernelGGL(image_scale_kernel, dim3((dst_w + 255) / 256, dst_h), dim3(256), 0, (hipStream_t)stream, d_src,
                       src_w, src_h, d_dst, dst_w, dst_h);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_center_image_placement(aur_ctx* ctx, const double* d_verts, int n_vertices, int verts_static,
                                          const double* d_energy, const double* d_phase, int n_frames, int img_w,
                                          int img_h, int zoom_percent, int center_motion, int32_t* d_placement,
                                          void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_center_image_placement: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(n_vertices >= 1 && img_w >= 1 && img_h >= 1, "aur_center_image_placement: bad size");
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_verts && d_energy && d_phase && d_placement, "aur_center_image_placement: null device pointer");
    const double user_zoom = fmax(0.1, (double)zoom_percent / 100.0);                     // rt_widget.py:1070
    const double cm = (double)center_motion / 100.0;                                      // :1073
    hipLaunchKernelGGL(image_placement_kernel, dim3(n_frames), dim3(64), 0, (hipStream_t)stream, d_verts, n_vertices,
                       verts_static ? 1 : 0, d_energy, d_phase, img_w, img_h, user_zoom, cm, d_placement);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_disc_composite(aur_ctx* ctx, const double* d_verts, int n_vertices, int verts_static, int n_frames,
                                  int W, int H, int box_x, int box_y, int box_w, int box_h, const aur_disc_source* src,
                                  int skip_full, uint32_t* d_full_plane, uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_disc_composite: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_disc_composite: frame size %dx%d unsupported", W, H);
    AUR_REQUIRE(n_vertices >= 3 && n_vertices <= 65536, "aur_disc_composite: %d vertices outside [3, 65536]", n_vertices);
    AUR_REQUIRE(box_x >= 0 && box_y >= 0 && box_w >= 1 && box_h >= 1 && box_x + box_w <= W && box_y + box_h <= H,
                "aur_disc_composite: box %d,%d %dx%d outside the %dx%d frame", box_x, box_y, box_w, box_h, W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_verts != nullptr && (d_frames != nullptr || d_full_plane != nullptr), "aur_disc_composite: null device pointer");
    DiscArgs A = {};
    A.verts = d_verts;
    A.nv = n_vertices;
    A.verts_static = verts_static ? 1 : 0;
    A.W = W; A.H = H; A.W32 = (W + 31) / 32;
    A.bx0 = box_x; A.by0 = box_y; A.bw = box_w; A.bh = box_h;
    A.skip_full = skip_full ? 1 : 0;
    A.full_plane = d_full_plane;
    A.frames = d_frames;
    if (d_frames != nullptr) {
        AUR_REQUIRE(src != nullptr, "aur_disc_composite: null source");
        AUR_REQUIRE(src->mode >= 0 && src->mode <= 2, "aur_disc_composite: source mode %d", src->mode);
        A.mode = src->mode;
        A.colour = src->colour_argb_pm;
        if (src->mode != 0) {
            AUR_REQUIRE(src->d_image && src->d_placement, "aur_disc_composite: image source without image or placement");
            AUR_REQUIRE(src->img_w >= 1 && src->img_h >= 1 && src->img_w <= 32767 && src->img_h <= 32767,
                        "aur_disc_composite: image %dx%d outside [1..32767]^2", src->img_w, src->img_h);
            A.image = src->d_image;
            A.img_w = src->img_w;
            A.img_h = src->img_h;
            A.placement = src->d_placement;
            A.placement_static = src->placement_static ? 1 : 0;
        }
    }
    // rows per workgroup: as many (at most 4, one per wave) as keep the accumulators within 80 KB of LDS
    int R = 4;
    while (R > 1 && (size_t)R * (2 * (size_t)box_w + 1) * 4 > 80 * 1024) R >>= 1;
    const size_t lds = (size_t)R * (2 * (size_t)box_w + 1) * 4;
    AUR_REQUIRE(lds <= 150 * 1024, "aur_disc_composite: box width %d too large", box_w);
    A.R = R;
    if (lds > 64 * 1024)
        AUR_CHECK_HIP(hipFuncSetAttribute((const void*)disc_aa_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(disc_aa_kernel, dim3((box_h + R - 1) / R, n_frames), dim3(256), lds, (hipStream_t)stream, A);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_compose_background(aur_ctx* ctx, const uint32_t* d_image, int img_w, int img_h, int tile, int x, int y,
                                      int dim_alpha, int W, int H, uint8_t* d_base, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_compose_background: frame size %dx%d unsupported", W, H);
    AUR_REQUIRE(img_w >= 1 && img_h >= 1 && img_w <= 32767 && img_h <= 32767, "aur_compose_background: image %dx%d outside [1..32767]^2",
                img_w, img_h);
    AUR_REQUIRE(dim_alpha >= 0 && dim_alpha <= 255, "aur_compose_background: dim alpha %d", dim_alpha);
    AUR_REQUIRE(x > -(1 << 24) && x < (1 << 24) && y > -(1 << 24) && y < (1 << 24), "aur_compose_background: offset out of range");
    AUR_REQUIRE(d_image && d_base, "aur_compose_background: null device pointer");
    hipLaunchKernelGGL(background_kernel, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, d_image, img_w,
                       img_h, tile ? 1 : 0, x, y, dim_alpha, W, H, d_base);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_blend_pixels(aur_ctx* ctx, const uint32_t* d_pixel_index, const uint8_t* d_coverage, int n_pixels,
                                uint32_t colour_argb_pm, int n_frames, int W, int H, uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_blend_pixels: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_blend_pixels: frame size %dx%d unsupported", W, H);
    AUR_REQUIRE(n_pixels >= 0, "aur_blend_pixels: n_pixels %d < 0", n_pixels);
    if (n_frames == 0 || n_pixels == 0) return AUR_OK;
    AUR_REQUIRE(d_pixel_index && d_coverage && d_frames, "aur_blend_pixels: null device pointer");
    hipLaunchKernelGGL(blend_pixels_kernel, dim3((n_pixels + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)stream,
                       d_pixel_index, d_coverage, n_pixels, colour_argb_pm >> 24, (colour_argb_pm >> 16) & 255u,
                       (colour_argb_pm >> 8) & 255u, colour_argb_pm & 255u, (long)W * H * 3, d_frames);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}
