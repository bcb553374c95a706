// Rasteriser for gfx950 (MI355X): per-frame column spans for the three linear export modes and the
// tile-per-workgroup RGB24 fill that turns spans into the frames handed to FFmpeg.
//
// Replaces widgets/rt_widget.py:821-852 (paint_frame: fillRect + mode dispatch), :934-940/:982-988
// (_draw_spectrum, linear), :903-913/:932 (_draw_waveform, linear), :990-1000 (_draw_particles) and the
// QImage -> numpy copy of export/exporter.py:84-95.
//
// Coverage rule (documented in oracle/raster_ref.py; Qt's own raster engine is not part of the reference
// tree, so pixel coverage is "parity unpinned" against Qt): a pixel is lit iff its centre lies inside the
// outline of the width-2 SquareCap/BevelJoin stroke.  Every frame is described by column spans
// (uint32 [F, K, W], r0 | r1 << 16 = lit rows [r0, r1) of a pixel column); the span kernels evaluate the
// reference's arithmetic in its own precision chain (float32 normalise -> float64 geometry -> int truncation),
// with contraction off, so they are bit-identical to the numpy oracle.  The fill kernel is pure HBM write
// bandwidth: each lane owns 4 pixels of a row (12 B, one dwordx3 store, contiguous across the wave) and walks
// the rows of its tile with its columns' spans held in registers.
#include "common.h"

#include <mutex>

namespace {

constexpr double kBig = 1.0e6;

__device__ __forceinline__ uint32_t pack_rows(long r0, long r1, int H) {
    if (r0 < 0) r0 = 0;
    if (r0 > H) r0 = H;
    if (r1 < 0) r1 = 0;
    if (r1 > H) r1 = H;
    if (r1 <= r0) return 0u;
    return (uint32_t)r0 | ((uint32_t)r1 << 16);
}

__device__ __forceinline__ float block_max_first(const float* row, int n, float* s_red) {
    // max(row[:n]) for n <= 128 by the first 128 threads of the block, result broadcast through LDS
    const int tid = threadIdx.x;
    float v = (tid < n) ? row[tid] : -INFINITY;
    if (tid < 128) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
        if ((tid & 63) == 0) s_red[tid >> 6] = v;
    }
    __syncthreads();
    return fmaxf(s_red[0], s_red[1]);
}

// ---- Spectrum - Linear: 80 bars (rt_widget.py:934-940, :982-988) ------------------------------------------
__global__ __launch_bounds__(256) void spans_spectrum_kernel(const float* __restrict__ logspec,
                                                             const double* __restrict__ energy, int W, int H,
                                                             uint32_t* __restrict__ spans) {
    __shared__ float s_red[2];
    __shared__ int s_bh[AUR_N_BARS];
    const int f = blockIdx.y;
    const float* spec = logspec + (long)f * AUR_NBINS;
    const float mxv = block_max_first(spec, AUR_N_BARS, s_red);
    const float mx = (mxv > 0.0f) ? mxv : 1.0f;                 // rt_widget.py:939
    const double scale = 0.4 + 0.6 * energy[f];                 // rt_widget.py:986
    if (threadIdx.x < AUR_N_BARS) {
        const float v32 = __fdiv_rn(spec[threadIdx.x], mx);    // spec / mx in float32 (rt_widget.py:940)
        const double v = (double)v32;
        s_bh[threadIdx.x] = (int)(v * (double)(H - 10) * scale);   // int(v * (h - 10) * (0.4 + 0.6 * energy))
    }
    __syncthreads();
    const int bar_w = (W / AUR_N_BARS) > 1 ? (W / AUR_N_BARS) : 1;   // max(1, int(w / N)), rt_widget.py:983
    uint32_t* out = spans + (long)f * AUR_K_SPECTRUM * W;
    for (int px = blockIdx.x * 256 + threadIdx.x; px < W; px += gridDim.x * 256) {
        uint32_t s0 = 0, s1 = 0;
        // drawLine(x0, h-5, x0, h-5-bh), width 2: columns {x0-1, x0}, rows [h-6-bh, h-4)
        if (px % bar_w == 0 && px / bar_w < AUR_N_BARS) {
            const int bh = s_bh[px / bar_w];
            const int ya = H - 5, yb = H - 5 - bh;
            s0 = pack_rows((ya < yb ? ya : yb) - 1, (ya > yb ? ya : yb) + 1, H);
        }
        if ((px + 1) % bar_w == 0 && (px + 1) / bar_w < AUR_N_BARS) {
            const int bh = s_bh[(px + 1) / bar_w];
            const int ya = H - 5, yb = H - 5 - bh;
            s1 = pack_rows((ya < yb ? ya : yb) - 1, (ya > yb ? ya : yb) + 1, H);
        }
        out[px] = s0;
        out[W + px] = s1;
    }
}

// ---- particles: 50 points (rt_widget.py:990-1000) ------------------------------------------------------------
__global__ __launch_bounds__(256) void spans_particles_kernel(const float* __restrict__ logspec, int W, int H,
                                                              uint32_t* __restrict__ spans) {
    __shared__ float s_red[2];
    __shared__ int s_x[AUR_N_PARTICLES], s_y[AUR_N_PARTICLES];
    const int f = blockIdx.y;
    const float* spec = logspec + (long)f * AUR_NBINS;
    const float mxv = block_max_first(spec, AUR_N_PARTICLES, s_red);
    const float den = mxv + 1e-6f;                              // np.max(spec) + 1e-6 stays float32 (NEP 50)
    if (threadIdx.x < AUR_N_PARTICLES) {
        const int i = threadIdx.x;
        const double v = (double)__fdiv_rn(spec[i], den);
        s_x[i] = (int)(((double)i + 0.5) / (double)AUR_N_PARTICLES * (double)W);
        s_y[i] = (int)((double)H / 2.0 + (v - 0.5) * (double)H * 0.6);
    }
    __syncthreads();
    uint32_t* out = spans + (long)f * AUR_K_PARTICLES * W;
    for (int px = blockIdx.x * 256 + threadIdx.x; px < W; px += gridDim.x * 256) {
        uint32_t s0 = 0, s1 = 0;
        // drawPoint(x, y), width 2: the 2x2 block {x-1, x} x {y-1, y}; later points overwrite earlier ones
        for (int i = 0; i < AUR_N_PARTICLES; ++i) {
            if (s_x[i] == px) s0 = pack_rows(s_y[i] - 1, s_y[i] + 1, H);
            if (s_x[i] - 1 == px) s1 = pack_rows(s_y[i] - 1, s_y[i] + 1, H);
        }
        out[px] = s0;
        out[W + px] = s1;
    }
}

// ---- Waveform - Linear: width-2 polyline (rt_widget.py:903-913, :932) ----------------------------------------

__device__ __forceinline__ uint32_t rows_from_interval(double ylo, double yhi, bool valid, int H) {
    if (!valid) return 0u;
    ylo = fmin(fmax(ylo, -kBig), kBig);
    yhi = fmin(fmax(yhi, -kBig), kBig);
    const long r0 = (long)ceil(ylo - 0.5);
    const long r1 = (long)floor(yhi - 0.5) + 1;
    return pack_rows(r0, r1, H);
}

// y-interval of segment `seg`'s stroke rectangle (half-width 1, SquareCap at the path ends) on x = px + 0.5
__device__ __forceinline__ uint32_t segment_rows(const double* Y, int seg, int px, int W, int H) {
    const int nseg = W - 1;
    if (seg < 0 || seg >= nseg) return 0u;
    const double y0 = Y[seg], y1 = Y[seg + 1];
    const double dy = y1 - y0;
    const double L2 = 1.0 + dy * dy;
    const double L = sqrt(L2);
    const double rx = ((double)px + 0.5) - (double)seg;
    const double c = rx * dy;
    const double tlo = c - L, thi = c + L;
    const double s0L = (seg == 0) ? -L : 0.0;
    const double s1L = (seg == nseg - 1) ? L2 + L : L2;
    double slo, shi;
    bool ok = true;
    if (dy > 0.0) {
        slo = (s0L - rx) / dy;
        shi = (s1L - rx) / dy;
    } else if (dy < 0.0) {
        slo = (s1L - rx) / dy;
        shi = (s0L - rx) / dy;
    } else {
        slo = -kBig;
        shi = kBig;
        ok = (s0L <= rx) && (rx <= s1L);
    }
    const double lo = fmax(tlo, slo), hi = fmin(thi, shi);
    return rows_from_interval(y0 + lo, y0 + hi, ok, H);
}

// y-interval of the BevelJoin triangle at interior vertex `v` on x = px + 0.5
__device__ __forceinline__ uint32_t bevel_rows(const double* Y, int v, int px, int W, int H) {
    if (v < 1 || v > W - 2) return 0u;
    const double yv = Y[v];
    const double dyA = yv - Y[v - 1];
    const double dyB = Y[v + 1] - yv;
    const double LA = sqrt(1.0 + dyA * dyA);
    const double LB = sqrt(1.0 + dyB * dyB);
    const double cr = dyB - dyA;
    if (cr == 0.0) return 0u;
    const double sg = (cr > 0.0) ? -1.0 : 1.0;
    const double u1 = (sg * (-dyA)) / LA, w1 = sg / LA;
    const double u2 = (sg * (-dyB)) / LB, w2 = sg / LB;
    const double uc = ((double)px + 0.5) - (double)v;
    double ymin = kBig, ymax = -kBig;
    bool hit = false;
    const double ua[3] = {0.0, 0.0, u1}, ya[3] = {0.0, 0.0, w1};
    const double ub[3] = {u1, u2, u2}, yb[3] = {w1, w2, w2};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const double lo_u = fmin(ua[e], ub[e]), hi_u = fmax(ua[e], ub[e]);
        if (ua[e] != ub[e] && lo_u <= uc && uc <= hi_u) {
            const double yc = ya[e] + ((uc - ua[e]) * (yb[e] - ya[e])) / (ub[e] - ua[e]);
            if (yc < ymin) ymin = yc;
            if (yc > ymax) ymax = yc;
            hit = true;
        }
    }
    return rows_from_interval(yv + ymin, yv + ymax, hit, H);
}

constexpr int kWaveCols = 256;   // columns per block (one per thread); Y staged with a 2-column apron

__global__ __launch_bounds__(256) void spans_waveform_kernel(const float* __restrict__ samples,
                                                             const int* __restrict__ col2samp,
                                                             const double* __restrict__ energy, int W, int H,
                                                             uint32_t* __restrict__ spans) {
    __shared__ double s_y[kWaveCols + 4];
    const int f = blockIdx.y;
    const float* smp = samples + (long)f * AUR_BLOCK;
    const double e = energy[f];
    const double amp = 0.45 * ((double)H / 2.0) * (0.5 + e);    // rt_widget.py:904
    const float amp32 = (float)amp;                             // np.float32 * python float -> float32 (NEP 50)
    const float mid32 = (float)((double)H / 2.0);               // rt_widget.py:905
    const int x0 = blockIdx.x * kWaveCols;
    for (int i = threadIdx.x; i < kWaveCols + 4; i += 256) {
        const int x = x0 - 2 + i;
        double y = 0.0;
        if (x >= 0 && x < W) y = (double)(mid32 - smp[col2samp[x]] * amp32);   // rt_widget.py:908-909
        s_y[i] = y;
    }
    __syncthreads();
    const int px = x0 + threadIdx.x;
    if (px >= W) return;
    // s_y index of column x is x - x0 + 2; pass a pointer whose index is the absolute column
    const double* Y = s_y + 2 - x0;
    uint32_t* out = spans + (long)f * AUR_K_WAVEFORM * W;
    uint32_t r[AUR_K_WAVEFORM] = {0u, 0u, 0u, 0u, 0u};
    if (W >= 2) {
        r[0] = segment_rows(Y, px - 1, px, W, H);
        r[1] = segment_rows(Y, px, px, W, H);
        r[2] = segment_rows(Y, px + 1, px, W, H);
        r[3] = bevel_rows(Y, px, px, W, H);
        r[4] = bevel_rows(Y, px + 1, px, W, H);
    }
#pragma unroll
    for (int k = 0; k < AUR_K_WAVEFORM; ++k) out[(long)k * W + px] = r[k];
}

// ---- fill: spans -> RGB24 frames ----------------------------------------------------------------------------
//
// Block = 256 lanes = 1024 pixel columns x kRowsPerBlock rows of one frame.  A lane loads the spans of its 4
// columns once (K x uint4), then for each row builds three dwords (4 RGB pixels) and stores them with one
// 12-byte store; consecutive lanes store consecutive 12-byte groups, so a wave writes 768 contiguous bytes per
// instruction.

constexpr int kRowsPerBlock = 16;
constexpr uint32_t kBg24 = 13u | (13u << 8) | (18u << 16);   // QColor(13, 13, 18), rt_widget.py:829

struct __attribute__((packed, aligned(4))) rgb4 {
    uint32_t d0, d1, d2;
};

template <int K>
__global__ __launch_bounds__(256) void raster_fill_kernel(const uint32_t* __restrict__ spans,
                                                          const uint32_t* __restrict__ pen_rgb, int W, int H,
                                                          uint8_t* __restrict__ frames) {
    const int f = blockIdx.z;
    const int px0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (px0 >= W) return;
    const int row0 = blockIdx.y * kRowsPerBlock;
    const int row1 = (row0 + kRowsPerBlock < H) ? row0 + kRowsPerBlock : H;
    const uint32_t pen = pen_rgb[f] & 0x00ffffffu;
    const uint32_t* sp = spans + (long)f * K * W + px0;

    uint32_t lo[K][4], len[K][4];
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint4 s = *reinterpret_cast<const uint4*>(sp + (long)k * W);
        const uint32_t v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lo[k][c] = v[c] & 0xffffu;
            len[k][c] = (v[c] >> 16) - (v[c] & 0xffffu);   // 0 for an empty span
            any |= v[c];
        }
    }
    uint8_t* base = frames + ((long)f * H + row0) * (long)W * 3 + (long)px0 * 3;
    const rgb4 bgv = {kBg24 | (kBg24 << 24), (kBg24 >> 8) | (kBg24 << 16), (kBg24 >> 16) | (kBg24 << 8)};
    if (any == 0) {
        for (int row = row0; row < row1; ++row) {
            *reinterpret_cast<rgb4*>(base) = bgv;
            base += (long)W * 3;
        }
        return;
    }
    for (int row = row0; row < row1; ++row) {
        uint32_t col[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bool lit = false;
#pragma unroll
            for (int k = 0; k < K; ++k) lit |= ((uint32_t)row - lo[k][c]) < len[k][c];
            col[c] = lit ? pen : kBg24;
        }
        rgb4 o;
        o.d0 = col[0] | (col[1] << 24);
        o.d1 = (col[1] >> 8) | (col[2] << 16);
        o.d2 = (col[2] >> 16) | (col[3] << 8);
        *reinterpret_cast<rgb4*>(base) = o;
        base += (long)W * 3;
    }
}

// Any width (the reference's UI offers every integer 64..3840, ui/main_window.py:147): one lane per pixel, three byte
// stores.  Nothing is assumed about alignment or divisibility; it is the slow path of odd sizes (854x480, 101x77).
template <int K>
__global__ __launch_bounds__(256) void raster_fill_px_kernel(const uint32_t* __restrict__ spans,
                                                             const uint32_t* __restrict__ pen_rgb, int W, int H,
                                                             const uint8_t* __restrict__ base,
                                                             uint8_t* __restrict__ frames) {
    const int f = blockIdx.z, row = blockIdx.y;
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= W) return;
    const uint32_t* sp = spans + (long)f * K * W + px;
    bool lit = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t v = sp[(long)k * W];
        lit |= ((uint32_t)row - (v & 0xffffu)) < ((v >> 16) - (v & 0xffffu));
    }
    uint32_t c = kBg24;
    if (base != nullptr) {          // aur_set_background: the frame starts from the composed background image
        const uint8_t* b = base + ((long)row * W + px) * 3;
        c = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
    }
    if (lit) c = pen_rgb[f] & 0x00ffffffu;
    uint8_t* dst = frames + (((long)f * H + row) * (long)W + px) * 3;
    dst[0] = (uint8_t)c;
    dst[1] = (uint8_t)(c >> 8);
    dst[2] = (uint8_t)(c >> 16);
}

// Fast path (W % 16 == 0, 16-byte aligned frames: every standard video size): the frame is walked as uint4
// (16-byte) columns.  16 bytes hold 5 1/3 RGB pixels, so uint4 column i of a row starts at pixel (16 i) / 3 with
// byte phase i % 3 and touches 6 pixel columns.  One wave owns 64 consecutive uint4 columns x kRowsPerTask rows:
// every store instruction writes 1 KB of contiguous, line-aligned bytes (full 128-byte lines, no partial-line
// traffic).  A lane keeps the spans of its 6 pixel columns in registers; rows outside the union of those spans
// (almost all of a frame) store a precomputed background pattern.
constexpr int kRowsPerTask = 16;

__device__ __forceinline__ uint4 pack_rgb16(const uint32_t (&P)[6], int ph) {
    const uint32_t s0 = P[0] | (P[1] << 24);
    const uint32_t s1 = (P[1] >> 8) | (P[2] << 16);
    const uint32_t s2 = (P[2] >> 16) | (P[3] << 8);
    const uint32_t s3 = P[4] | (P[5] << 24);
    const uint32_t s4 = P[5] >> 8;
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(s1, s0, ph);
    o.y = __builtin_amdgcn_alignbyte(s2, s1, ph);
    o.z = __builtin_amdgcn_alignbyte(s3, s2, ph);
    o.w = __builtin_amdgcn_alignbyte(s4, s3, ph);
    return o;
}

template <int K, bool kNT>
__global__ __launch_bounds__(256) void raster_fill16_kernel(const uint32_t* __restrict__ spans,
                                                            const uint32_t* __restrict__ pen_rgb, int W, int H,
                                                            int nu4, int nchunks, int rows_per_task,
                                                            uint4* __restrict__ frames) {
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int chunk = task % nchunks, tile = task / nchunks;
    const int row0 = tile * rows_per_task;
    const int i = chunk * 64 + lane;
    if (row0 >= H || i >= nu4) return;
    const int row1 = (row0 + rows_per_task < H) ? row0 + rows_per_task : H;
    const int c0 = (16 * i) / 3;
    const int ph = i % 3;                               // (16 i) mod 3
    const uint32_t pen = pen_rgb[f] & 0x00ffffffu;
    const uint32_t* sp = spans + (long)f * K * W;

    uint32_t lo[K][6], len[K][6];
    uint32_t umin = 0xffffu, umax = 0u;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const uint32_t v = sp[(long)k * W + c0 + c];
            const uint32_t a = v & 0xffffu, b = v >> 16;
            lo[k][c] = a;
            len[k][c] = b - a;                          // 0 for an empty span
            if (b > a) {
                umin = a < umin ? a : umin;
                umax = b > umax ? b : umax;
            }
        }
    }
    const uint32_t ulen = umax > umin ? umax - umin : 0u;
    const uint32_t Pbg[6] = {kBg24, kBg24, kBg24, kBg24, kBg24, kBg24};
    const uint4 bgv = pack_rgb16(Pbg, ph);
    uint4* dst = frames + ((long)f * H + row0) * (long)nu4 + i;
    for (int row = row0; row < row1; ++row) {
        uint4 o = bgv;
        if (((uint32_t)row - umin) < ulen) {
            uint32_t P[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                bool lit = false;
#pragma unroll
                for (int k = 0; k < K; ++k) lit |= ((uint32_t)row - lo[k][c]) < len[k][c];
                P[c] = lit ? pen : kBg24;
            }
            o = pack_rgb16(P, ph);
        }
        if (kNT) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 ov = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(dst));
        } else {
            *dst = o;
        }
        dst += nu4;
    }
}

// ---- linear fill: one aligned 4 KB piece per workgroup, one store per lane -----------------------------------
//
// Measured on MI355X (tools/fill_lab.hip): a pure fill reaches 6.9 TB/s only when every 256-lane workgroup writes ONE
// 4 KB-aligned 4 KB piece with one 16-byte store per lane and workgroups walk the buffer in address order; 16 rows x
// 1 KB per wave (raster_fill16_kernel) tops out at 5.6 TB/s because its pieces are neither 1 KB- nor 4 KB-aligned
// (a 4K row is 11.25 KB) and more than one store per lane costs 5-15 %.  A lane therefore owns exactly one uint4 of
// the whole batch [F * H * nu4], and what it must know about its six pixel columns comes from ONE 8-byte look-up in a
// per-frame column summary built by span_summary_kernel (a second, dependent look-up only next to a polyline).
//
// Summary of uint4 column `col` of frame f (pixel columns c0 .. c0 + 5, c0 = 16 col / 3):
//   .x = umin | (umax - umin) << 16   union of the lit rows of the six pixel columns (0: nothing lit)
//   .y = mask6 | flags                bit c: pixel column c0 + c has a span; kSumUniform: every span of the six columns
//                                     is the same interval (bars, dots): lit pixels = mask6 inside the union;
//                                     kSumMerged: each pixel column's spans form one interval, stored in merged[f][x]
constexpr uint32_t kSumUniform = 1u << 8;
constexpr uint32_t kSumMerged = 1u << 9;

template <int K>
__global__ __launch_bounds__(256) void span_summary_kernel(const uint32_t* __restrict__ spans, int W, int nu4,
                                                           int n_frames, uint2* __restrict__ summary,
                                                           uint32_t* __restrict__ merged) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_frames * nu4) return;
    const int f = idx / nu4, col = idx - f * nu4;
    const int c0 = (16 * col) / 3;
    const uint32_t* sp = spans + (long)f * K * W + c0;
    uint32_t umin = 0xffffu, umax = 0u, mask = 0u;
    uint32_t fa = 0u, fb = 0u;                   // first non-empty interval
    bool uniform = true, all_merged = true;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        uint32_t a[K], b[K];
        uint32_t lo = 0xffffu, hi = 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t v = sp[(long)k * W + c];
            a[k] = v & 0xffffu;
            b[k] = v >> 16;
            if (b[k] > a[k]) {
                lo = a[k] < lo ? a[k] : lo;
                hi = b[k] > hi ? b[k] : hi;
                if (fb == 0u) {
                    fa = a[k];
                    fb = b[k];
                } else if (a[k] != fa || b[k] != fb) {
                    uniform = false;
                }
            }
        }
        uint32_t m = 0u;
        if (hi > lo) {
            mask |= 1u << c;
            umin = lo < umin ? lo : umin;
            umax = hi > umax ? hi : umax;
            // contiguous iff growing from `lo` through overlapping / touching intervals reaches `hi`
            uint32_t reach = lo;
#pragma unroll
            for (int pass = 0; pass < K; ++pass)
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (b[k] > a[k] && a[k] <= reach && b[k] > reach) reach = b[k];
            if (reach == hi) m = lo | (hi << 16);
            else all_merged = false;
        }
        merged[(long)f * W + c0 + c] = m;       // neighbouring uint4 columns share pixel columns: identical values
    }
    uint2 s;
    s.x = umax > umin ? (umin | ((umax - umin) << 16)) : 0u;
    s.y = mask | (uniform ? kSumUniform : 0u) | (all_merged ? kSumMerged : 0u);
    summary[idx] = s;
}

// BASE: the frame starts from `base` (uint4 [upf], aur_set_background) instead of the plain (13,13,18) pattern: one more
// 16-byte load per lane (the base frame is shared by every frame of the batch and stays in the last-level cache).
template <int K, bool BASE>
__global__ __launch_bounds__(256) void raster_fill_lin_kernel(const uint32_t* __restrict__ spans,
                                                              const uint2* __restrict__ summary,
                                                              const uint32_t* __restrict__ merged,
                                                              const uint32_t* __restrict__ pen_rgb, int W, int H,
                                                              int nu4, int upf, float inv_upf, float inv_nu4,
                                                              int head, unsigned total,
                                                              const uint4* __restrict__ base,
                                                              uint4* __restrict__ frames) {
    // g = index of this lane's uint4 in the batch; frames + g is 4 KB aligned for lane 0 of every workgroup
    const unsigned gi = blockIdx.x * 256u + threadIdx.x;
    if (gi < (unsigned)head) return;
    const unsigned g = gi - (unsigned)head;
    if (g >= total) return;
    // f = g / upf, row = (g % upf) / nu4 by float reciprocal + one-step fix-up (quotients are small: exact after it)
    int f = (int)((float)g * inv_upf);
    int u = (int)(g - (unsigned)f * (unsigned)upf);
    if (u < 0) {
        u += upf;
        --f;
    } else if (u >= upf) {
        u -= upf;
        ++f;
    }
    int row = (int)((float)u * inv_nu4);
    int col = u - row * nu4;
    if (col < 0) {
        col += nu4;
        --row;
    } else if (col >= nu4) {
        col -= nu4;
        ++row;
    }
    const uint2 s = summary[f * nu4 + col];
    const uint32_t pen = pen_rgb[f] & 0x00ffffffu;
    const int ph = col % 3;                             // (16 col) mod 3
    // background (13,13,18): dword j of the 16 bytes starts at byte (ph + j) % 3 of the repeating pattern 0d 0d 12
    constexpr uint32_t D0 = kBg24 | (kBg24 << 24), D1 = (kBg24 >> 8) | (kBg24 << 16), D2 = (kBg24 >> 16) | (kBg24 << 8);
    uint4 o;
    if (BASE) {
        o = base[u];
    } else {
        o.x = ph == 0 ? D0 : (ph == 1 ? D1 : D2);
        o.y = ph == 0 ? D1 : (ph == 1 ? D2 : D0);
        o.z = ph == 0 ? D2 : (ph == 1 ? D0 : D1);
        o.w = o.x;
    }
    if (((uint32_t)row - (s.x & 0xffffu)) < (s.x >> 16)) {
        uint32_t lb = 0u;                                 // bit c: pixel column c0 + c is lit in this row
        if (s.y & kSumUniform) {
            lb = s.y & 0x3fu;
        } else if (s.y & kSumMerged) {
            const uint32_t* m = merged + (long)f * W + (16 * col) / 3;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const uint32_t v = m[c];
                lb |= ((((uint32_t)row - (v & 0xffffu)) < ((v >> 16) - (v & 0xffffu))) ? 1u : 0u) << c;
            }
        } else {
            const uint32_t* sp = spans + (long)f * K * W + (16 * col) / 3;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                bool lit = false;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const uint32_t v = sp[(long)k * W + c];
                    lit |= ((uint32_t)row - (v & 0xffffu)) < ((v >> 16) - (v & 0xffffu));
                }
                lb |= (lit ? 1u : 0u) << c;
            }
        }
        uint32_t P[6];
        if (BASE) {
            uint32_t M[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                M[c] = ((lb >> c) & 1u) ? 0x00ffffffu : 0u;
                P[c] = pen & M[c];
            }
            const uint4 mk = pack_rgb16(M, ph), pv = pack_rgb16(P, ph);
            o.x = (o.x & ~mk.x) | pv.x;
            o.y = (o.y & ~mk.y) | pv.y;
            o.z = (o.z & ~mk.z) | pv.z;
            o.w = (o.w & ~mk.w) | pv.w;
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) P[c] = ((lb >> c) & 1u) ? pen : kBg24;
            o = pack_rgb16(P, ph);
        }
    }
    aur_store_frame16(frames + g, o);
}

int check_raster_dims(const char* who, int n_frames, int W, int H) {
    AUR_REQUIRE(n_frames >= 0, "%s: n_frames %d < 0", who, n_frames);
    // W >= 64: the reference UI's smallest export width (ui/main_window.py:147); from 50 columns on no two particles share
    // a column, which the two span slots per column rely on
    AUR_REQUIRE(W >= 64 && W <= 16384 && H >= 16 && H <= 16384, "%s: frame size %dx%d outside [64..16384]x[16..16384]",
                who, W, H);
    AUR_REQUIRE(n_frames <= 65535, "%s: n_frames %d > 65535 per call", who, n_frames);
    return AUR_OK;
}

}  // namespace

extern "C" int aur_spans_spectrum_linear(aur_ctx* ctx, const float* d_logspec, const double* d_energy, int n_frames,
                                         int W, int H, uint32_t* d_spans, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_raster_dims("aur_spans_spectrum_linear", n_frames, W, H))) return rc;
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_logspec && d_energy && d_spans, "aur_spans_spectrum_linear: null device pointer");
    const unsigned gx = (unsigned)((W + 255) / 256);
    hipLaunchKernelGGL(spans_spectrum_kernel, dim3(gx, n_frames), dim3(256), 0, (hipStream_t)stream, d_logspec,
                       d_energy, W, H, d_spans);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_spans_particles(aur_ctx* ctx, const float* d_logspec, int n_frames, int W, int H,
                                   uint32_t* d_spans, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_raster_dims("aur_spans_particles", n_frames, W, H))) return rc;
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_logspec && d_spans, "aur_spans_particles: null device pointer");
    const unsigned gx = (unsigned)((W + 255) / 256);
    hipLaunchKernelGGL(spans_particles_kernel, dim3(gx, n_frames), dim3(256), 0, (hipStream_t)stream, d_logspec, W,
                       H, d_spans);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_spans_waveform_linear(aur_ctx* ctx, const float* d_samples, const int32_t* d_col2samp,
                                         const double* d_energy, int n_frames, int W, int H, uint32_t* d_spans,
                                         void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_raster_dims("aur_spans_waveform_linear", n_frames, W, H))) return rc;
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_samples && d_col2samp && d_energy && d_spans, "aur_spans_waveform_linear: null device pointer");
    const unsigned gx = (unsigned)((W + kWaveCols - 1) / kWaveCols);
    hipLaunchKernelGGL(spans_waveform_kernel, dim3(gx, n_frames), dim3(256), 0, (hipStream_t)stream, d_samples,
                       d_col2samp, d_energy, W, H, d_spans);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

// Work area of the linear fill: uint2 [F, nu4] column summaries, then uint32 [F, W] merged intervals.
static size_t raster_work_bytes(int n_frames, int W) {
    return (size_t)n_frames * (size_t)(W * 3 / 16) * 8 + (size_t)n_frames * (size_t)W * 4;
}

// Shapes the linear fill takes: 16-byte columns, at least one wave of them per row, 32-bit batch indices.
static bool raster_lin_ok(int n_frames, int W, int H, const void* d_frames) {
    if (W % 16 != 0 || ((uintptr_t)d_frames & 15u) != 0) return false;
    const long upf = (long)(W * 3 / 16) * H;
    return upf >= 256 && upf < (1L << 26) && (long)n_frames * upf + 256 < (1L << 31);
}

extern "C" int aur_raster_work_bytes(int n_frames, int W, size_t* out_bytes) {
    AUR_REQUIRE(out_bytes != nullptr && n_frames >= 0 && W >= 16 && W <= 16384, "aur_raster_work_bytes: bad argument");
    *out_bytes = raster_work_bytes(n_frames, W);
    return AUR_OK;
}

extern "C" int aur_span_summary(aur_ctx* ctx, const uint32_t* d_spans, int K, int n_frames, int W, int H, void* d_work,
                                size_t work_bytes, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_raster_dims("aur_span_summary", n_frames, W, H))) return rc;
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_spans && d_work, "aur_span_summary: null device pointer");
    AUR_REQUIRE(K == 2 || K == 5, "aur_span_summary: K=%d unsupported (2 or 5)", K);
    AUR_REQUIRE(W % 16 == 0, "aur_span_summary: width %d must be a multiple of 16", W);
    AUR_REQUIRE(work_bytes >= raster_work_bytes(n_frames, W), "aur_span_summary: work area of %zu bytes, need %zu",
                work_bytes, raster_work_bytes(n_frames, W));
    AUR_REQUIRE(((uintptr_t)d_work & 15u) == 0 && ((uintptr_t)d_spans & 3u) == 0, "aur_span_summary: misaligned pointer");
    const int nu4 = W * 3 / 16;
    uint2* summary = reinterpret_cast<uint2*>(d_work);
    uint32_t* merged = reinterpret_cast<uint32_t*>(summary + (size_t)n_frames * nu4);
    const unsigned g = (unsigned)(((long)n_frames * nu4 + 255) / 256);
    if (K == 2)
        hipLaunchKernelGGL(span_summary_kernel<2>, dim3(g), dim3(256), 0, (hipStream_t)stream, d_spans, W, nu4, n_frames,
                           summary, merged);
    else
        hipLaunchKernelGGL(span_summary_kernel<5>, dim3(g), dim3(256), 0, (hipStream_t)stream, d_spans, W, nu4, n_frames,
                           summary, merged);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_raster_spans_ws(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb, int n_frames,
                                   int W, int H, uint8_t* d_frames, void* d_work, size_t work_bytes, int work_ready,
                                   void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_raster_dims("aur_raster_spans_ws", n_frames, W, H))) return rc;
    if (d_work == nullptr || ctx->fill_mode == 1 || !raster_lin_ok(n_frames, W, H, d_frames))
        return aur_raster_spans(ctx, d_spans, K, d_pen_rgb, n_frames, W, H, d_frames, stream);
    if (n_frames == 0) return AUR_OK;
    if ((rc = aur_check_background(ctx, "aur_raster_spans_ws", W, H))) return rc;
    AUR_REQUIRE(d_spans && d_pen_rgb && d_frames, "aur_raster_spans_ws: null device pointer");
    AUR_REQUIRE(K == 2 || K == 5, "aur_raster_spans_ws: K=%d unsupported (2 or 5)", K);
    AUR_REQUIRE(((uintptr_t)d_pen_rgb & 3u) == 0, "aur_raster_spans_ws: misaligned device pointer");
    if (!work_ready) {
        if ((rc = aur_span_summary(ctx, d_spans, K, n_frames, W, H, d_work, work_bytes, stream))) return rc;
    } else {
        AUR_REQUIRE(work_bytes >= raster_work_bytes(n_frames, W) && ((uintptr_t)d_work & 15u) == 0,
                    "aur_raster_spans_ws: work area too small or misaligned");
    }
    const int nu4 = W * 3 / 16, upf = nu4 * H;
    const uint2* summary = reinterpret_cast<const uint2*>(d_work);
    const uint32_t* merged = reinterpret_cast<const uint32_t*>(summary + (size_t)n_frames * nu4);
    const uint32_t* pen = reinterpret_cast<const uint32_t*>(d_pen_rgb);
    hipStream_t st = (hipStream_t)stream;
    const int head = (int)(((uintptr_t)d_frames & 4095u) / 16u);       // uint4s between the 4 KB boundary and frame 0
    const unsigned total = (unsigned)n_frames * (unsigned)upf;
    const unsigned gfill = (unsigned)(((unsigned long)total + (unsigned)head + 255u) / 256u);
    uint4* out = reinterpret_cast<uint4*>(d_frames);                   // lane 0 of workgroup b stores at out + 256 b - head
    const uint4* base = reinterpret_cast<const uint4*>(ctx->d_base);   // 16-byte aligned (aur_set_background)
#define AUR_LAUNCH_LIN(KK, BB)                                                                                          \
    hipLaunchKernelGGL((raster_fill_lin_kernel<KK, BB>), dim3(gfill), dim3(256), 0, st, d_spans, summary, merged, pen, W, \
                       H, nu4, upf, 1.0f / (float)upf, 1.0f / (float)nu4, head, total, base, out)
    if (K == 2) {
        if (base) AUR_LAUNCH_LIN(2, true); else AUR_LAUNCH_LIN(2, false);
    } else {
        if (base) AUR_LAUNCH_LIN(5, true); else AUR_LAUNCH_LIN(5, false);
    }
#undef AUR_LAUNCH_LIN
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

// ---- allocation-free variant of aur_raster_spans_ws ------------------------------------------------------------------
//
// The work area of the linear fill (column summaries + merged intervals) lives in a file-scope device array instead of
// caller memory, for callers that cannot allocate (HIP graph capture, serving loops).  Every GPU the library runs on
// holds its own copy of the array at its own address: the address is resolved once per device in aur_create
// (aur_raster_static_init, with that device current) and kept in the context; host code never uses the array's name for
// anything but that look-up.
static __device__ __align__(256) unsigned char scratch[AUR_STATIC_WORK_BYTES];

// Every call on a GPU shares that GPU's array.  Calls on one stream are ordered by the stream; a call on ANOTHER stream
// while the previous call's kernels may still be running would rewrite summaries those kernels read.  The library keeps,
// per device, the stream of the last call and an event recorded behind it (created in aur_create, never in the entry
// point), and refuses such a call instead of racing: hipEventQuery, no allocation, no synchronisation.  Streams under
// graph capture are exempt on both sides (an event recorded there is a graph node and cannot be queried): launches
// inside one captured graph are ordered by the graph's own edges, which is the caller's business.
constexpr int kMaxGuardDevices = 64;
struct StaticGuard {
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool pending = false;
};
static StaticGuard g_guard[kMaxGuardDevices];
static std::mutex g_guard_mu;

int aur_raster_static_init(aur_ctx* ctx) {
    void* p = nullptr;
    AUR_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));      // the context's device is current (aur_create)
    AUR_REQUIRE(p != nullptr && ((uintptr_t)p & 255u) == 0, "aur_create: static work area unresolved or misaligned");
    ctx->d_static_work = p;
    if (ctx->device >= 0 && ctx->device < kMaxGuardDevices) {
        std::lock_guard<std::mutex> lock(g_guard_mu);
        if (g_guard[ctx->device].done == nullptr)
            AUR_CHECK_HIP(hipEventCreateWithFlags(&g_guard[ctx->device].done, hipEventDisableTiming));
    }
    return AUR_OK;
}

extern "C" int aur_raster_spans_static(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb,
                                       int n_frames, int W, int H, uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_raster_dims("aur_raster_spans_static", n_frames, W, H))) return rc;
    AUR_REQUIRE(ctx->d_static_work != nullptr, "aur_raster_spans_static: static work area not initialised");
    if (ctx->fill_mode == 1 || !raster_lin_ok(n_frames, W, H, d_frames))      // shapes of the row-tile fill: no work area
        return aur_raster_spans(ctx, d_spans, K, d_pen_rgb, n_frames, W, H, d_frames, stream);
    const size_t need = raster_work_bytes(n_frames, W);
    AUR_REQUIRE(need <= (size_t)AUR_STATIC_WORK_BYTES,
                "aur_raster_spans_static: %d frames of width %d need a work area of %zu bytes, the static one holds %d "
                "(split the batch: at most %d frames of this width per call)",
                n_frames, W, need, AUR_STATIC_WORK_BYTES,
                (int)((size_t)AUR_STATIC_WORK_BYTES / raster_work_bytes(1, W)));
    hipStream_t st = (hipStream_t)stream;
    StaticGuard* g = (ctx->device >= 0 && ctx->device < kMaxGuardDevices) ? &g_guard[ctx->device] : nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (g != nullptr && hipStreamIsCapturing(st, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    const bool guarded = g != nullptr && g->done != nullptr && cap == hipStreamCaptureStatusNone;
    std::unique_lock<std::mutex> lock(g_guard_mu, std::defer_lock);
    if (guarded) {
        lock.lock();                            // held over the launch: two threads cannot both pass the check
        if (g->pending && g->last != st) {
            const hipError_t q = hipEventQuery(g->done);
            if (q == hipErrorNotReady) {
                aur_set_error("aur_raster_spans_static: the previous call on this GPU (stream %p) has not finished and this "
                              "call is on another stream (%p): both would use the one static work area; order the streams "
                              "(event / synchronise) or use aur_raster_spans_ws with a work area per stream",
                              (void*)g->last, (void*)st);
                return AUR_EINVAL;
            }
            if (q != hipSuccess) (void)hipGetLastError();
            g->pending = false;
        }
    }
    // the summary kernel rewrites every element the fill reads for this shape (summaries of all n_frames * nu4 columns,
    // merged intervals of all n_frames * W pixel columns), so nothing of the previous call's contents is ever read
    rc = aur_raster_spans_ws(ctx, d_spans, K, d_pen_rgb, n_frames, W, H, d_frames, ctx->d_static_work,
                             (size_t)AUR_STATIC_WORK_BYTES, 0, stream);
    if (guarded && rc == AUR_OK && n_frames > 0) {
        AUR_CHECK_HIP(hipEventRecord(g->done, st));
        g->last = st;
        g->pending = true;
    }
    return rc;
}

extern "C" int aur_raster_spans(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb, int n_frames,
                                int W, int H, uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_raster_dims("aur_raster_spans", n_frames, W, H))) return rc;
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_spans && d_pen_rgb && d_frames, "aur_raster_spans: null device pointer");
    AUR_REQUIRE(((uintptr_t)d_spans & 3u) == 0 && ((uintptr_t)d_pen_rgb & 3u) == 0,
                "aur_raster_spans: misaligned device pointer");
    const uint32_t* pen = reinterpret_cast<const uint32_t*>(d_pen_rgb);
    hipStream_t st = (hipStream_t)stream;
    if (K != 2 && K != 5) {
        aur_set_error("aur_raster_spans: K=%d unsupported (2 or 5)", K);
        return AUR_EINVAL;
    }
    if ((rc = aur_check_background(ctx, "aur_raster_spans", W, H))) return rc;
    // any width, any alignment -- and every frame that starts from a background image outside the linear fill's shapes
    if (ctx->d_base != nullptr || W % 4 != 0 || ((uintptr_t)d_frames & 3u) != 0 || ((uintptr_t)d_spans & 15u) != 0) {
        const dim3 grid((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)n_frames);
        if (K == 2)
            hipLaunchKernelGGL(raster_fill_px_kernel<2>, grid, dim3(256), 0, st, d_spans, pen, W, H, ctx->d_base, d_frames);
        else
            hipLaunchKernelGGL(raster_fill_px_kernel<5>, grid, dim3(256), 0, st, d_spans, pen, W, H, ctx->d_base, d_frames);
        AUR_CHECK_HIP(hipGetLastError());
        return AUR_OK;
    }
    if (W % 16 == 0 && ((uintptr_t)d_frames & 15u) == 0) {
        const int nu4 = W * 3 / 16, nchunks = (nu4 + 63) / 64;
        const int rpt = ctx->fill_rows > 0 ? ctx->fill_rows : kRowsPerTask;
        const int ntiles = (H + rpt - 1) / rpt;
        const dim3 grid((unsigned)((nchunks * ntiles + 3) / 4), (unsigned)n_frames);
        uint4* out = reinterpret_cast<uint4*>(d_frames);
#define AUR_LAUNCH_FILL16(KK, NT)                                                                              \
    hipLaunchKernelGGL((raster_fill16_kernel<KK, NT>), grid, dim3(256), 0, st, d_spans, pen, W, H, nu4, nchunks, \
                       rpt, out)
        if (K == 2) {
            if (ctx->fill_nt) AUR_LAUNCH_FILL16(2, true); else AUR_LAUNCH_FILL16(2, false);
        } else {
            if (ctx->fill_nt) AUR_LAUNCH_FILL16(5, true); else AUR_LAUNCH_FILL16(5, false);
        }
#undef AUR_LAUNCH_FILL16
    } else {   // widths that are multiples of 4: 12-byte stores
        const dim3 grid((unsigned)((W / 4 + 255) / 256), (unsigned)((H + kRowsPerBlock - 1) / kRowsPerBlock),
                        (unsigned)n_frames);
        if (K == 2)
            hipLaunchKernelGGL(raster_fill_kernel<2>, grid, dim3(256), 0, st, d_spans, pen, W, H, d_frames);
        else
            hipLaunchKernelGGL(raster_fill_kernel<5>, grid, dim3(256), 0, st, d_spans, pen, W, H, d_frames);
    }
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}
