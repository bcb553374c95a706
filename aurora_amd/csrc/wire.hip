// Wire format of the frames handed to FFmpeg (export/exporter.py:429-434: "-f rawvideo -pix_fmt rgb24 -s WxH").
//
// The reference pipes packed rgb24 (3 B per pixel) and lets FFmpeg's swscale convert to yuv420p for x264
// (exporter.py:485-490 "-pix_fmt yuv420p").  Every exported frame crosses PCIe, which bounds the end-to-end rate far
// below the HBM rate of the rasteriser, so the conversion can run on the GPU instead: planar yuv420p is 1.5 B per pixel,
// half the readback bytes, and FFmpeg is then started with "-pix_fmt yuv420p" for its input.
// This is opt-in (view_state["hip_wire_format"] = "yuv420p"): the default wire format stays the reference's rgb24.
// "nv12" is the same conversion with the chroma interleaved (one Cb Cr plane), the layout the reference asks FFmpeg's
// swscale for in front of the VAAPI encoders (exporter.py:459-466 "-vf format=nv12,hwupload ... -pix_fmt nv12").
//
// Conversion (stated in full because FFmpeg is not available to pin it; oracle/wire_ref.py restates it in numpy):
// ITU-R BT.601, limited range, 8-bit integer form
//     Y  = (( 66 R + 129 G +  25 B + 128) >> 8) + 16
//     Cb = ((-38 R -  74 G + 112 B + 128) >> 8) + 128
//     Cr = ((112 R -  94 G -  18 B + 128) >> 8) + 128          (>> is an arithmetic shift)
// with Cb, Cr computed from the rounded mean (sum + 2) >> 2 of each 2x2 pixel block.  Bound: HBM (3 B read + 1.5 B
// written per pixel, no reuse).
#include "common.h"

namespace {

// Chroma of one lane's two 2x2 blocks (u, v: two bytes each) of row pair y2.  Planar (yuv420p): a Cb plane, then a Cr
// plane, W/2 bytes per row.  Semi-planar (nv12): one plane of interleaved Cb Cr pairs, W bytes per row.
template <bool NV12>
__device__ __forceinline__ void store_chroma(uint8_t* dst, size_t frame_px, int W, int y2, int x4, uint32_t u, uint32_t v) {
    if (NV12) {
        const uint32_t uv = (u & 255u) | ((v & 255u) << 8) | ((u >> 8) << 16) | ((v >> 8) << 24);
        *reinterpret_cast<uint32_t*>(dst + frame_px + (size_t)y2 * W + 4 * x4) = uv;
    } else {
        const size_t coff = (size_t)y2 * (W / 2) + 2 * x4;
        *reinterpret_cast<uint16_t*>(dst + frame_px + coff) = (uint16_t)u;
        *reinterpret_cast<uint16_t*>(dst + frame_px + frame_px / 4 + coff) = (uint16_t)v;
    }
}

// One lane: 4 pixels x 2 rows.  A wave reads 768 contiguous bytes of each of the two rows (12 B per lane, dwordx3),
// writes 256 contiguous bytes of each Y row and 128 contiguous bytes of the U and of the V row.
template <bool NV12>
__global__ __launch_bounds__(256) void rgb24_to_yuv420p_kernel(const uint8_t* __restrict__ rgb, int W, int H,
                                                               uint8_t* __restrict__ yuv) {
    const int f = blockIdx.z;
    const int y2 = blockIdx.y;                                   // row pair
    const int x4 = blockIdx.x * 256 + threadIdx.x;               // group of 4 pixels
    if (x4 * 4 >= W) return;
    const size_t frame_px = (size_t)W * H;
    const uint8_t* src = rgb + (size_t)f * frame_px * 3 + ((size_t)(2 * y2) * W + 4 * x4) * 3;
    uint8_t* dst = yuv + (size_t)f * (frame_px + frame_px / 2);
    struct __attribute__((aligned(4))) u3 {
        uint32_t a, b, c;
    };
    const u3 r0 = *reinterpret_cast<const u3*>(src);
    const u3 r1 = *reinterpret_cast<const u3*>(src + (size_t)W * 3);
    uint32_t yw[2];
    int sr[2] = {0, 0}, sg[2] = {0, 0}, sb[2] = {0, 0};         // 2x2 block sums, blocks 0 and 1 of this lane
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        const u3 q = row ? r1 : r0;
        const uint32_t w[3] = {q.a, q.b, q.c};
        uint32_t out = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            auto byte_at = [&](int i) { return (int)((w[i >> 2] >> (8 * (i & 3))) & 0xffu); };
            const int R = byte_at(3 * p), G = byte_at(3 * p + 1), B = byte_at(3 * p + 2);
            const int Y = ((66 * R + 129 * G + 25 * B + 128) >> 8) + 16;
            out |= (uint32_t)Y << (8 * p);
            sr[p >> 1] += R;
            sg[p >> 1] += G;
            sb[p >> 1] += B;
        }
        yw[row] = out;
    }
    uint32_t* ydst = reinterpret_cast<uint32_t*>(dst + (size_t)(2 * y2) * W + 4 * x4);
    ydst[0] = yw[0];
    *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(ydst) + W) = yw[1];
    uint32_t u = 0, v = 0;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int R = (sr[b] + 2) >> 2, G = (sg[b] + 2) >> 2, B = (sb[b] + 2) >> 2;
        const int Cb = ((-38 * R - 74 * G + 112 * B + 128) >> 8) + 128;
        const int Cr = ((112 * R - 94 * G - 18 * B + 128) >> 8) + 128;
        u |= (uint32_t)Cb << (8 * b);
        v |= (uint32_t)Cr << (8 * b);
    }
    store_chroma<NV12>(dst, frame_px, W, y2, x4, u, v);
}

// ---- spans -> yuv420p directly (the span modes with the opt-in wire format) -------------------------------------------
// A frame of the span modes holds two colours, background and pen, so its planar form follows from the spans alone: the
// rgb24 frame (3 B written, 3 B read per pixel by the converter above) is never materialised; 1.5 B per pixel are written
// once.  Same arithmetic as rgb24_to_yuv420p_kernel on the frame aur_raster_spans would have produced (bit-identical:
// tests/test_gpu_parity.py::test_fused_yuv420p_fill).  One lane: 4 pixel columns, kPairs row pairs, spans in registers.
constexpr int kPairs = 8;
constexpr int kBgR = 13, kBgG = 13, kBgB = 18;                  // QColor(13, 13, 18), rt_widget.py:829

template <int K, bool NV12>
__global__ __launch_bounds__(256) void spans_to_yuv420p_kernel(const uint32_t* __restrict__ spans,
                                                               const uint32_t* __restrict__ pen_rgb, int W, int H,
                                                               uint8_t* __restrict__ yuv) {
    const int f = blockIdx.z;
    const int x4 = blockIdx.x * 256 + threadIdx.x;
    if (x4 * 4 >= W) return;
    const int pair0 = blockIdx.y * kPairs;
    const uint32_t pen = pen_rgb[f];
    const int pr = (int)(pen & 255u), pg = (int)((pen >> 8) & 255u), pb = (int)((pen >> 16) & 255u);
    const uint32_t y_bg = (uint32_t)(((66 * kBgR + 129 * kBgG + 25 * kBgB + 128) >> 8) + 16);
    const uint32_t y_pen = (uint32_t)(((66 * pr + 129 * pg + 25 * pb + 128) >> 8) + 16);
    uint32_t lo[K][4], len[K][4];
    const uint32_t* sp = spans + (long)f * K * W + 4 * x4;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint4 s = *reinterpret_cast<const uint4*>(sp + (long)k * W);
        const uint32_t v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lo[k][c] = v[c] & 0xffffu;
            len[k][c] = (v[c] >> 16) - (v[c] & 0xffffu);
        }
    }
    const size_t frame_px = (size_t)W * H;
    uint8_t* dst = yuv + (size_t)f * (frame_px + frame_px / 2);
    for (int pr2 = pair0; pr2 < pair0 + kPairs && 2 * pr2 < H; ++pr2) {
        int cnt[2] = {0, 0};                                    // lit pixels of the lane's two 2x2 blocks
        uint32_t yw[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t row = (uint32_t)(2 * pr2 + r);
            uint32_t out = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bool lit = false;
#pragma unroll
                for (int k = 0; k < K; ++k) lit |= (row - lo[k][c]) < len[k][c];
                out |= (lit ? y_pen : y_bg) << (8 * c);
                cnt[c >> 1] += lit ? 1 : 0;
            }
            yw[r] = out;
        }
        uint32_t* ydst = reinterpret_cast<uint32_t*>(dst + (size_t)(2 * pr2) * W + 4 * x4);
        ydst[0] = yw[0];
        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(ydst) + W) = yw[1];
        uint32_t u = 0, v = 0;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = cnt[b];
            const int R = (n * pr + (4 - n) * kBgR + 2) >> 2, G = (n * pg + (4 - n) * kBgG + 2) >> 2;
            const int B = (n * pb + (4 - n) * kBgB + 2) >> 2;
            const int Cb = ((-38 * R - 74 * G + 112 * B + 128) >> 8) + 128;
            const int Cr = ((112 * R - 94 * G - 18 * B + 128) >> 8) + 128;
            u |= (uint32_t)Cb << (8 * b);
            v |= (uint32_t)Cr << (8 * b);
        }
        store_chroma<NV12>(dst, frame_px, W, pr2, x4, u, v);
    }
}

template <bool NV12>
int raster_spans_planar(const char* who, aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb,
                        int n_frames, int W, int H, uint8_t* d_yuv, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "%s: n_frames %d outside [0, 65535]", who, n_frames);
    AUR_REQUIRE(W >= 64 && W <= 16384 && H >= 16 && H <= 16384, "%s: frame size %dx%d unsupported", who, W, H);
    AUR_REQUIRE(W % 4 == 0 && H % 2 == 0, "%s: %dx%d: width must be a multiple of 4, height even", who, W, H);
    AUR_REQUIRE(K == 2 || K == 5, "%s: K=%d unsupported (2 or 5)", who, K);
    AUR_REQUIRE(ctx->d_base == nullptr, "%s: a background image is set; render rgb24 and convert (aur_rgb24_to_%s)", who,
                NV12 ? "nv12" : "yuv420p");
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_spans && d_pen_rgb && d_yuv, "%s: null device pointer", who);
    AUR_REQUIRE(((uintptr_t)d_spans & 15u) == 0 && ((uintptr_t)d_pen_rgb & 3u) == 0 && ((uintptr_t)d_yuv & 3u) == 0,
                "%s: misaligned device pointer", who);
    const dim3 grid((W / 4 + 255) / 256, (H / 2 + kPairs - 1) / kPairs, n_frames);
    const uint32_t* pen = reinterpret_cast<const uint32_t*>(d_pen_rgb);
    if (K == 2)
        hipLaunchKernelGGL((spans_to_yuv420p_kernel<2, NV12>), grid, dim3(256), 0, (hipStream_t)stream, d_spans, pen, W, H,
                           d_yuv);
    else
        hipLaunchKernelGGL((spans_to_yuv420p_kernel<5, NV12>), grid, dim3(256), 0, (hipStream_t)stream, d_spans, pen, W, H,
                           d_yuv);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

template <bool NV12>
int rgb24_to_planar(const char* who, aur_ctx* ctx, const uint8_t* d_rgb, int n_frames, int W, int H, uint8_t* d_yuv,
                    void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "%s: n_frames %d outside [0, 65535]", who, n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "%s: frame size %dx%d unsupported", who, W, H);
    AUR_REQUIRE(W % 4 == 0 && H % 2 == 0, "%s: %dx%d: width must be a multiple of 4, height even", who, W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_rgb && d_yuv, "%s: null device pointer", who);
    AUR_REQUIRE(((uintptr_t)d_rgb & 3u) == 0 && ((uintptr_t)d_yuv & 3u) == 0, "%s: device pointers must be 4-byte aligned",
                who);
    hipLaunchKernelGGL(rgb24_to_yuv420p_kernel<NV12>, dim3((W / 4 + 255) / 256, H / 2, n_frames), dim3(256), 0,
                       (hipStream_t)stream, d_rgb, W, H, d_yuv);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

}  // namespace

extern "C" int aur_raster_spans_yuv420p(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb,
                                        int n_frames, int W, int H, uint8_t* d_yuv, void* stream) {
    return raster_spans_planar<false>("aur_raster_spans_yuv420p", ctx, d_spans, K, d_pen_rgb, n_frames, W, H, d_yuv, stream);
}

extern "C" int aur_raster_spans_nv12(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb, int n_frames,
                                     int W, int H, uint8_t* d_nv12, void* stream) {
    return raster_spans_planar<true>("aur_raster_spans_nv12", ctx, d_spans, K, d_pen_rgb, n_frames, W, H, d_nv12, stream);
}

extern "C" int aur_rgb24_to_yuv420p(aur_ctx* ctx, const uint8_t* d_rgb, int n_frames, int W, int H, uint8_t* d_yuv,
                                    void* stream) {
    return rgb24_to_planar<false>("aur_rgb24_to_yuv420p", ctx, d_rgb, n_frames, W, H, d_yuv, stream);
}

extern "C" int aur_rgb24_to_nv12(aur_ctx* ctx, const uint8_t* d_rgb, int n_frames, int W, int H, uint8_t* d_nv12,
                                 void* stream) {
    return rgb24_to_planar<true>("aur_rgb24_to_nv12", ctx, d_rgb, n_frames, W, H, d_nv12, stream);
}
