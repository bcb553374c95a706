// Internal declarations shared by the translation units of libaurora_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/aurora_hip.h"

// Device-resident constant tables, computed in float64 on the host (abi.hip: build_tables).
//   d_window : float32[2048]  periodic Hann w[n]                          audio/analysis.py:9
//   d_twid   : float64, complex values stored as (re, im):
//     [AUR_TW_W1 ..)  w1[l]     = W1024^l              l < 64   pass-1 twiddle base (powers are built per frame)
//     [AUR_TW_WU ..)  wu[l]     = 0.5 * W2048^l        l < 64   real-FFT unpack twiddle base (0.5 folded in)
//     [AUR_TW_W2 ..)  tw2[r][q] = W64^(r*q)            r < 4, q < 16   pass-2 twiddles
//     [AUR_TW_WS ..)  ws[l]     = wu[l] for l > 0, 0.5 * W2048^128 for l = 0   unpack base of lane 0's shifted pairs
//     [AUR_TW_P1 ..)  p1[k-1][l] = W1024^(l*k)         k = 1..15, l < 64   pass-1 twiddles
//     [AUR_TW_LG ..)  lg[i]     = (1 / c_i, -log(1 / c_i)), c_i = 1 + i / 128, i <= 128: table of the correctly rounded
//                                 log1p of the displayed bins (stft.hip: log1p_cr); not complex values
//   with W_N = exp(-2 pi i / N).
#define AUR_TW_W1      0
#define AUR_TW_WU      128
#define AUR_TW_W2      256
#define AUR_TW_WS      384
#define AUR_TW_P1      512
#define AUR_TW_LG      (512 + 15 * 128)
#define AUR_LG_ENTRIES 129
#define AUR_TW_DOUBLES (AUR_TW_LG + 2 * AUR_LG_ENTRIES)

struct aur_ctx {
    int device;
    float* d_window;       // AUR_FFT_SIZE floats
    double* d_twid;        // AUR_TW_DOUBLES doubles
    float h_window[AUR_FFT_SIZE];
    int num_cus;
    // tuning knobs, read once from the environment in aur_create (0 = built-in default)
    int fill_rows;         // AUR_FILL_ROWS: rows per wave task of the fill kernel
    int fill_nt;           // AUR_FILL_NT=1: non-temporal frame stores
    int stft_zt;           // AUR_STFT_ZT=0: hop <= 896 keeps the general STFT kernel instead of the zero-tail variant (A/B runs)
    int fill_mode;         // AUR_FILL_MODE=1: aur_raster_spans_ws / aur_raster_planes use the row-tile kernels (A/B runs)
    int stft_fpw;          // AUR_STFT_FPW: frames per wave of the STFT kernel
    void* d_static_work;   // this device's address of raster.hip's file-scope work area (aur_raster_spans_static)
    const uint8_t* d_base; // aur_set_background: uint8 [base_h][base_w][3] every frame starts from (NULL: plain (13,13,18))
    int base_w, base_h;
    void* d_rows;          // AUR_ROWS_BYTES of per-(frame, row) disc intervals + AUR_ROWS_FLAGS per-frame flags: the small
                           // table aur_raster_planes' linear fill looks up (radial.hip); allocated in aur_create
};

#define AUR_ROWS_BYTES (8 * 1024 * 1024)
#define AUR_ROWS_FLAGS 65536

// abi.hip: a background is set for exactly this frame size (or none is set)
int aur_check_background(aur_ctx* ctx, const char* who, int W, int H);

// raster.hip: resolves the device address of the static work area for ctx->device (which must be current)
int aur_raster_static_init(aur_ctx* ctx);

void aur_set_error(const char* fmt, ...);

#define AUR_CHECK_HIP(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            aur_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return AUR_EHIP;                                                                \
        }                                                                                   \
    } while (0)

#define AUR_REQUIRE(cond, ...)                                                              \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            aur_set_error(__VA_ARGS__);                                                     \
            return AUR_EINVAL;                                                              \
        }                                                                                   \
    } while (0)

// Sets the context's device current for the calling thread.
static inline int aur_use_device(aur_ctx* ctx) {
    if (!ctx) {
        aur_set_error("null context");
        return AUR_EINVAL;
    }
    AUR_CHECK_HIP(hipSetDevice(ctx->device));
    return AUR_OK;
}

// Frame stores of the fill kernels.  Finished frames are written once and not read again by the kernel that writes
// them, so they are stored non-temporally (global_store_dwordx4 ... nt): measured on MI355X the linear fill goes from
// 6.46 to 6.85 TB/s (profiles/r01_fill_nt.txt).  -DAUR_STREAM_STORES=0 builds the plain-store variant for A/B runs.
#ifndef AUR_STREAM_STORES
#define AUR_STREAM_STORES 1
#endif
__device__ __forceinline__ void aur_store_frame16(uint4* p, const uint4& v) {
#if AUR_STREAM_STORES
    typedef unsigned int aur_u32x4 __attribute__((ext_vector_type(4)));
    const aur_u32x4 ov = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(ov, reinterpret_cast<aur_u32x4*>(p));
#else
    *p = v;
#endif
}
