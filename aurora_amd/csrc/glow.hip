// Glow decoration of the closed paths (gfx950): `_draw_glow_path` -> `_draw_glow_offscreen`, widgets/rt_widget.py:598-616,
// :634-653 -- the path stroked once per layer with an antialiased RoundCap / RoundJoin pen of growing width and falling
// alpha, CompositionMode_Plus.
//
// The layers are nested strokes of ONE path, so one pass finds, for every pixel near the path, the distance to it and the
// nearest point; every layer's coverage follows from that distance (oracle/raster_ref.py: path_distance_field,
// round_stroke_coverage, plus_cov -- the rule is "parity unpinned" against Qt's stroker / gray raster; this file matches
// the oracle bit for bit: float64, same operation order, contraction off).  Replaces the earlier seven stroke bit planes
// (one atomicOr pass per layer) + aur_add_planes, whose coverage was aliased.
//
// Search structure, rebuilt per batch in the work area: the path's segments in groups of 16 consecutive ones with their
// bounding boxes grown by the reach of the widest layer.  A workgroup owns a 64 x 64 pixel region: 256 lanes test the
// (at most 256) group boxes against it; regions without a hit -- almost all of a frame -- end there.  Each wave then
// walks 8 x 8 pixel tiles, one pixel per lane, narrows the region's groups down to the tile's and scans their segments.
#include "common.h"
// A/B build knob of tools/glow_lesion.sh (where the kernel's time goes): 1 = no layer loop, 2 = one segment per group,
// 3 = region test only, 4 = empty kernel
#ifndef GLOW_LESION
#define GLOW_LESION 0
#endif

#include <math.h>

namespace {

constexpr int kMaxGroup = 16;          // segments per group: 4, 8 or 16 -- the smallest that keeps a path within kMaxGroups
constexpr int kMaxGroups = 256;        // n_vertices <= 4096
constexpr int kMaxLayers = 16;         // rt_widget.py:635: layers = max(4, min(16, glow_layers))
constexpr double kReach = 0.7072;      // oracle GLOW_REACH

struct GlowArgs {
    const double* segs;        // [F][nseg][5] = ax, ay, ex, ey, 1 / |e|^2
    const float4* boxes;       // [F][ngroups] = x0, y0, x1, y1 (pixel-centre coordinates a group can matter for)
    int nseg, ngroups, group, W, H, W32, K;
    double hw[kMaxLayers];     // half pen width of every layer, in paint order
    double hw_max;
    const uint32_t* pm;        // [F][K] premultiplied 0x00BBGGRR (bytes r, g, b, 0) per frame and layer
    const uint32_t* excl_a;    // [F][H][W32] or NULL: pixels an opaque layer painted afterwards covers (the path's stroke)
    const uint32_t* excl_b;    // [F or 1][H][W32] or NULL (the fully covered part of the disc)
    int excl_b_static;
    uint8_t* frames;
};

__global__ __launch_bounds__(256) void glow_prepare_kernel(const double* __restrict__ verts, int nv, double reach,
                                                           double* __restrict__ segs, float4* __restrict__ boxes,
                                                           int ngroups, int kGroup) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double* V = verts + (long)f * nv * 2;
    if (i < nv) {
        const int i1 = (i + 1 == nv) ? 0 : i + 1;
        double ax = V[2 * i], ay = V[2 * i + 1], bx = V[2 * i1], by = V[2 * i1 + 1];
        double ex = bx - ax, ey = by - ay, inv = 0.0;
        if (!(ax == ax && ay == ay && bx == bx && by == by)) {     // NaN vertex: the segment is nowhere
            ax = ay = 1.0e30;
            ex = ey = 0.0;
        } else {
            const double l2 = ex * ex + ey * ey;
            inv = l2 > 0.0 ? 1.0 / l2 : 0.0;
        }
        double* s = segs + ((long)f * nv + i) * 5;
        s[0] = ax; s[1] = ay; s[2] = ex; s[3] = ey; s[4] = inv;
    }
    if (i < nv && (i % kGroup) == 0) {
        double x0 = 1.0e30, y0 = 1.0e30, x1 = -1.0e30, y1 = -1.0e30;
        for (int k = 0; k < kGroup && i + k < nv; ++k) {
            const int a = i + k, b = (a + 1 == nv) ? 0 : a + 1;
            const double ax = V[2 * a], ay = V[2 * a + 1], bx = V[2 * b], by = V[2 * b + 1];
            if (!(ax == ax && ay == ay && bx == bx && by == by)) continue;
            x0 = fmin(x0, fmin(ax, bx)); x1 = fmax(x1, fmax(ax, bx));
            y0 = fmin(y0, fmin(ay, by)); y1 = fmax(y1, fmax(ay, by));
        }
        // grown by the reach and rounded outwards to whole pixels (exact in float32)
        float4 bb;
        bb.x = (float)fmax(-1.0e6, floor(x0 - reach - 1.0));
        bb.y = (float)fmax(-1.0e6, floor(y0 - reach - 1.0));
        bb.z = (float)fmin(1.0e6, ceil(x1 + reach + 1.0));
        bb.w = (float)fmin(1.0e6, ceil(y1 + reach + 1.0));
        boxes[(long)f * ngroups + i / kGroup] = bb;
    }
}

// Area of the unit pixel on the inner side of the stroke's rim (oracle: round_stroke_coverage), 0..255; a >= b are the
// absolute components of the unit normal, largest first.
__device__ __forceinline__ uint32_t rim_coverage(double s, double a, double b, double inv_a, double inv_2ab) {
    if (s >= kReach) return 255u;
    if (s <= -kReach) return 0u;
    const double u = s + (a + b) / 2.0;
    double f;
    if (u <= 0.0) f = 0.0;
    else if (u >= a + b) f = 1.0;
    else if (u < b) f = (u * u) * inv_2ab;
    else if (u <= a) f = (u - b / 2.0) * inv_a;
    else {
        const double r = (a + b) - u;
        f = 1.0 - (r * r) * inv_2ab;
    }
    double c = floor(f * 256.0);
    if (!(c == c) || c > 1.0e300 || c < -1.0e300) c = 0.0;
    c = fmin(fmax(c, 0.0), 255.0);
    return (uint32_t)c;
}

// Nearest point of segment (ax, ay) + t (ex, ey), t in [0, 1]: a + clamp(t) e gives a and a + e exactly at the ends.
__device__ __forceinline__ double seg_dist2(const double* __restrict__ sg, double px, double py, double& qx, double& qy) {
    const double ax = sg[0], ay = sg[1], ex = sg[2], ey = sg[3], inv = sg[4];
    const double tt = ((px - ax) * ex + (py - ay) * ey) * inv;
    const double tc = fmin(fmax(tt, 0.0), 1.0);
    qx = ax + tc * ex;
    qy = ay + tc * ey;
    const double dx = px - qx, dy = py - qy;
    return dx * dx + dy * dy;
}

// Minimum over the 64 lanes as DPP row operations (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror,
// row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3; lane 63 holds the result).  Every lane must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_ones(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = umin32(v, dpp_ones<0xB1, 0xF>(v));
    v = umin32(v, dpp_ones<0x4E, 0xF>(v));
    v = umin32(v, dpp_ones<0x141, 0xF>(v));
    v = umin32(v, dpp_ones<0x140, 0xF>(v));
    v = umin32(v, dpp_ones<0x142, 0xA>(v));
    v = umin32(v, dpp_ones<0x143, 0xC>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int kGroup>
__global__ __launch_bounds__(256) void glow_kernel(GlowArgs A) {
    __shared__ int s_cand[kMaxGroups];
    __shared__ float4 s_box[kMaxGroups];       // boxes of the region's candidate groups, in s_cand order
    __shared__ int s_count[4];
    __shared__ double s_seg[4][kGroup * 5];
    const int f = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rx0 = blockIdx.x * 64, ry0 = blockIdx.y * 64;
#if GLOW_LESION == 4
    if (A.W > 0) return;
#endif
    const float4* boxes = A.boxes + (long)f * A.ngroups;
    // region test: pixel centres of the region span [rx0 + .5, rx0 + 63.5]
    bool hit = false;
    float4 my_box = {0.0f, 0.0f, 0.0f, 0.0f};
    if ((int)threadIdx.x < A.ngroups) {
        const float4 bb = boxes[threadIdx.x];
        my_box = bb;
        hit = bb.x <= (float)(rx0 + 64) && bb.z >= (float)rx0 && bb.y <= (float)(ry0 + 64) && bb.w >= (float)ry0;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_count[wave] = __popcll(m);
    __syncthreads();
    int base = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += s_count[w];
        total += s_count[w];
    }
    if (total == 0) return;                                        // nothing of the path reaches this region
#if GLOW_LESION == 3
    if (A.W > 0) return;
#endif
    if (hit) {
        const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
        s_cand[slot] = (int)threadIdx.x;
        s_box[slot] = my_box;
    }
    __syncthreads();
    const double* segs = A.segs + (long)f * A.nseg * 5;
    const uint32_t* pm = A.pm + (long)f * A.K;
    const double rmax = A.hw_max + kReach;
    // group boxes are grown by hw_max + 2 px; the true (ungrown) box is what bounds a group's distance from below
    const float grow = (float)(A.hw_max + 2.0);
    // the first 64 candidate groups (all of them, in all but pathological regions) stay in registers for every tile
    int g_first = 0;
    float4 bb_first = {0.0f, 0.0f, 0.0f, 0.0f};
    if (lane < total) {
        g_first = s_cand[lane];
        bb_first = s_box[lane];
    }
    for (int t = wave; t < 64; t += 4) {
        const int tx0 = rx0 + (t & 7) * 8, ty0 = ry0 + (t >> 3) * 8;
        if (tx0 >= A.W || ty0 >= A.H) continue;                    // wave-uniform
        const int x = tx0 + (lane & 7), y = ty0 + (lane >> 3);
        const double px = (double)x + 0.5, py = (double)y + 0.5;
        const float pxf = (float)px, pyf = (float)py;
        // nothing farther than rmax matters (its coverage is 0 in every layer): starting from that bound lets the tiles
        // that lie inside a group's box but outside the band skip the group without scanning it
        double best = rmax * rmax * 1.000001;
        int best_si = 0x7fffffff;
        for (int c0 = 0; c0 < total; c0 += 64) {
            bool th = false;
            int g = g_first;
            float gd = 0.0f;                                       // lower bound of the tile centre's distance to the group
            float4 bb = bb_first;
            if (c0 + lane < total) {
                if (c0 > 0) {                                      // regions with more than 64 candidate groups: rare
                    g = s_cand[c0 + lane];
                    bb = s_box[c0 + lane];
                }
                th = bb.x <= (float)(tx0 + 8) && bb.z >= (float)tx0 && bb.y <= (float)(ty0 + 8) && bb.w >= (float)ty0;
                const float cxf = (float)tx0 + 4.0f, cyf = (float)ty0 + 4.0f;
                const float ddx = fmaxf(fmaxf(bb.x - cxf, cxf - bb.z), 0.0f), ddy = fmaxf(fmaxf(bb.y - cyf, cyf - bb.w), 0.0f);
                gd = ddx * ddx + ddy * ddy;
            }
            if (__ballot(th) == 0ull) continue;
            // nearest group first, so that the others can be skipped against a good bound; the order does not change the
            // result: ties between segments go to the lower index explicitly.  The key is the distance (a float >= 0
            // orders like its bit pattern) with the lane in its low six bits: one wave minimum names the group's lane.
            uint32_t key = th ? ((__float_as_uint(gd) & 0xffffffc0u) | (uint32_t)lane) : 0xffffffffu;
            for (;;) {
                const uint32_t mn = wave_min_u32(key);
                if (mn == 0xffffffffu) break;
                const int bit = (int)(mn & 63u);
                if (lane == bit) key = 0xffffffffu;
                const int grp = __builtin_amdgcn_readlane(g, bit);
                // this lane's lower bound on the distance to anything in the group (ungrown box, float32 rounded down)
                const float bx0 = readlane_f(bb.x, bit) + grow, by0 = readlane_f(bb.y, bit) + grow;
                const float bx1 = readlane_f(bb.z, bit) - grow, by1 = readlane_f(bb.w, bit) - grow;
                const float ldx = fmaxf(fmaxf(bx0 - pxf, pxf - bx1), 0.0f), ldy = fmaxf(fmaxf(by0 - pyf, pyf - by1), 0.0f);
                const float lb = (ldx * ldx + ldy * ldy) * 0.999f - 1.0e-3f;
                if (__ballot((double)lb <= best) == 0ull) continue;   // no pixel of the tile can find a nearer point there
                // the group's 16 segments (80 doubles) go through a wave-private LDS strip: one coalesced load, then
                // broadcast reads -- a global load per segment and lane made every iteration wait for memory
                const int s0 = grp * kGroup;
                double* strip = s_seg[wave];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < kGroup * 5; i += 64) {
                    const int si = s0 + i / 5;
                    // past the end of the path: a segment that is nowhere
                    strip[i] = (si < A.nseg) ? segs[(long)s0 * 5 + i] : ((i % 5) < 2 ? 1.0e30 : 0.0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int j = 0; j < (GLOW_LESION == 2 ? 1 : kGroup); ++j) {
                    double qx, qy;
                    const double dd = seg_dist2(strip + 5 * j, px, py, qx, qy);
                    const int si = s0 + j;
                    if (dd < best || (dd == best && si < best_si)) {
                        best = dd;
                        best_si = si;
                    }
                }
            }
        }
        if (x >= A.W || y >= A.H) continue;
        const double d = sqrt(best);
        if (!(d <= rmax)) continue;
        const long wi = (long)y * A.W32 + (x >> 5);
        const uint32_t bitm = 1u << (x & 31);
        if (A.excl_a != nullptr && (A.excl_a[(long)f * A.H * A.W32 + wi] & bitm)) continue;
        if (A.excl_b != nullptr && (A.excl_b[(A.excl_b_static ? 0L : (long)f * A.H * A.W32) + wi] & bitm)) continue;
        double bqx, bqy;
        (void)seg_dist2(segs + (long)best_si * 5, px, py, bqx, bqy);
        // one reciprocal per pixel for the normal, two for the rim formula: shared by every layer (oracle: inv_d, inv_a,
        // inv_2ab -- the same operations in the same order)
        const double inv_d = 1.0 / d;
        const double nx = fabs((px - bqx) * inv_d), ny = fabs((py - bqy) * inv_d);
        const double na = fmax(nx, ny), nb = fmin(nx, ny);
        const double inv_a = 1.0 / na, inv_2ab = 1.0 / ((2.0 * na) * nb);
        uint8_t* p = A.frames + (((long)f * A.H + y) * A.W + x) * 3;
        uint32_t r = p[0], g8 = p[1], b8 = p[2];
        const uint32_t r_in = r, g_in = g8, b_in = b8;
#if GLOW_LESION == 1
        r = (uint32_t)(d + na + nb) & 255u;
        for (int k = 0; k < 0; ++k) {
#else
        for (int k = 0; k < A.K; ++k) {
#endif
            const uint32_t cv = rim_coverage(A.hw[k] - d, na, nb, inv_a, inv_2ab);
            if (cv == 0u) continue;
            const uint32_t c = pm[k];
            uint32_t ar = r + (c & 255u), ag = g8 + ((c >> 8) & 255u), ab = b8 + ((c >> 16) & 255u);
            ar = ar > 255u ? 255u : ar;
            ag = ag > 255u ? 255u : ag;
            ab = ab > 255u ? 255u : ab;
            const uint32_t ic = 255u - cv;
            uint32_t v = ar * cv + r * ic;
            r = (v + (v >> 8) + 0x80u) >> 8;
            v = ag * cv + g8 * ic;
            g8 = (v + (v >> 8) + 0x80u) >> 8;
            v = ab * cv + b8 * ic;
            b8 = (v + (v >> 8) + 0x80u) >> 8;
        }
        if (r != r_in) p[0] = (uint8_t)r;
        if (g8 != g_in) p[1] = (uint8_t)g8;
        if (b8 != b_in) p[2] = (uint8_t)b8;
    }
}

int glow_group_size(int nv) { return nv <= 4 * kMaxGroups ? 4 : (nv <= 8 * kMaxGroups ? 8 : 16); }

size_t glow_work_bytes(int n_frames, int nv) {
    const int kGroup = glow_group_size(nv);
    const size_t ngroups = (size_t)(nv + kGroup - 1) / kGroup;
    const size_t segs = (size_t)n_frames * nv * 5 * sizeof(double);
    return ((segs + 255) & ~(size_t)255) + (size_t)n_frames * ngroups * sizeof(float4);
}

}  // namespace

extern "C" int aur_glow_work_bytes(int n_frames, int n_vertices, size_t* out_bytes) {
    AUR_REQUIRE(out_bytes != nullptr && n_frames >= 0 && n_vertices >= 3 && n_vertices <= kMaxGroups * kMaxGroup,
                "aur_glow_work_bytes: bad argument");
    *out_bytes = glow_work_bytes(n_frames, n_vertices);
    return AUR_OK;
}

extern "C" int aur_glow_path(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H, int K,
                             const double* h_half_widths, const uint8_t* d_pm_rgb, const uint32_t* d_excl_a,
                             const uint32_t* d_excl_b, int excl_b_static, void* d_work, size_t work_bytes,
                             uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_glow_path: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_glow_path: frame size %dx%d unsupported", W, H);
    AUR_REQUIRE(n_vertices >= 3 && n_vertices <= kMaxGroups * kMaxGroup, "aur_glow_path: %d vertices outside [3, %d]",
                n_vertices, kMaxGroups * kMaxGroup);
    AUR_REQUIRE(K >= 1 && K <= kMaxLayers, "aur_glow_path: K %d outside [1, %d]", K, kMaxLayers);
    AUR_REQUIRE(h_half_widths != nullptr, "aur_glow_path: null width table");
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_verts && d_pm_rgb && d_work && d_frames, "aur_glow_path: null device pointer");
    AUR_REQUIRE(((uintptr_t)d_pm_rgb & 3u) == 0 && ((uintptr_t)d_work & 255u) == 0, "aur_glow_path: misaligned pointer");
    AUR_REQUIRE(work_bytes >= glow_work_bytes(n_frames, n_vertices), "aur_glow_path: work area of %zu bytes, need %zu",
                work_bytes, glow_work_bytes(n_frames, n_vertices));
    GlowArgs A = {};
    A.hw_max = 0.0;
    for (int k = 0; k < K; ++k) {
        AUR_REQUIRE(h_half_widths[k] > 0.0 && h_half_widths[k] <= 256.0, "aur_glow_path: half width %g outside (0, 256]",
                    h_half_widths[k]);
        A.hw[k] = h_half_widths[k];
        A.hw_max = fmax(A.hw_max, h_half_widths[k]);
    }
    const int kGroup = glow_group_size(n_vertices);
    const int ngroups = (n_vertices + kGroup - 1) / kGroup;
    double* segs = reinterpret_cast<double*>(d_work);
    const size_t seg_bytes = (((size_t)n_frames * n_vertices * 5 * sizeof(double)) + 255) & ~(size_t)255;
    float4* boxes = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(d_work) + seg_bytes);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(glow_prepare_kernel, dim3((n_vertices + 255) / 256, n_frames), dim3(256), 0, st, d_verts,
                       n_vertices, A.hw_max + 1.0, segs, boxes, ngroups, kGroup);
    A.segs = segs;
    A.boxes = boxes;
    A.nseg = n_vertices;
    A.ngroups = ngroups;
    A.group = kGroup;
    A.W = W; A.H = H; A.W32 = (W + 31) / 32;
    A.K = K;
    A.pm = reinterpret_cast<const uint32_t*>(d_pm_rgb);
    A.excl_a = d_excl_a;
    A.excl_b = d_excl_b;
    A.excl_b_static = excl_b_static ? 1 : 0;
    A.frames = d_frames;
    const dim3 grid((W + 63) / 64, (H + 63) / 64, n_frames);
    if (kGroup == 4) hipLaunchKernelGGL(glow_kernel<4>, grid, dim3(256), 0, st, A);
    else if (kGroup == 8) hipLaunchKernelGGL(glow_kernel<8>, grid, dim3(256), 0, st, A);
    else hipLaunchKernelGGL(glow_kernel<16>, grid, dim3(256), 0, st, A);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}
