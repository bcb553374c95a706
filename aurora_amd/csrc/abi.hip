// Context, tables, error reporting and host-memory helpers of the C ABI (include/aurora_hip.h).
#include "common.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void aur_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* aur_last_error(void) { return g_err; }
extern "C" int aur_version(void) { return 1; }

extern "C" int aur_device_count(int* out_count) {
    AUR_REQUIRE(out_count != nullptr, "aur_device_count: null out_count");
    AUR_CHECK_HIP(hipGetDeviceCount(out_count));
    return AUR_OK;
}

// Window (float64 -> float32) and float64 twiddle bases (layout: common.h).
static void build_tables(float* window, double* tw) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int m = 0; m < AUR_FFT_SIZE; ++m) {
        // scipy.signal.get_window('hann', N, fftbins=True) == 0.5 - 0.5 cos(2 pi m / N) (audio/analysis.py:9)
        window[m] = (float)(0.5 - 0.5 * cos(two_pi * (double)m / (double)AUR_FFT_SIZE));
    }
    for (int l = 0; l < 64; ++l) {
        const double a1 = -two_pi * (double)l / 1024.0;
        tw[AUR_TW_W1 + 2 * l] = cos(a1);
        tw[AUR_TW_W1 + 2 * l + 1] = sin(a1);
        const double au = -two_pi * (double)l / 2048.0;
        tw[AUR_TW_WU + 2 * l] = 0.5 * cos(au);
        tw[AUR_TW_WU + 2 * l + 1] = 0.5 * sin(au);
        const double as = (l == 0) ? -two_pi * 128.0 / 2048.0 : au;
        tw[AUR_TW_WS + 2 * l] = 0.5 * cos(as);
        tw[AUR_TW_WS + 2 * l + 1] = 0.5 * sin(as);
    }
    for (int k = 1; k < 16; ++k)
        for (int l = 0; l < 64; ++l) {
            const double a = -two_pi * (double)((l * k) % 1024) / 1024.0;
            tw[AUR_TW_P1 + 2 * ((k - 1) * 64 + l)] = cos(a);
            tw[AUR_TW_P1 + 2 * ((k - 1) * 64 + l) + 1] = sin(a);
        }
    for (int i = 0; i < AUR_LG_ENTRIES; ++i) {
        const double inv = 1.0 / (1.0 + (double)i / 128.0);
        tw[AUR_TW_LG + 2 * i] = inv;
        tw[AUR_TW_LG + 2 * i + 1] = (i == 0) ? 0.0 : (double)(-logl((long double)inv));
    }
    for (int r = 0; r < 4; ++r)
        for (int q = 0; q < 16; ++q) {
            const double a = -two_pi * (double)(r * q) / 64.0;
            tw[AUR_TW_W2 + 2 * (r * 16 + q)] = cos(a);
            tw[AUR_TW_W2 + 2 * (r * 16 + q) + 1] = sin(a);
        }
}

extern "C" int aur_device_pci_bus_id(int device, char* out, int len) {
    AUR_REQUIRE(out != nullptr && len >= 16, "aur_device_pci_bus_id: buffer of at least 16 bytes expected");
    AUR_CHECK_HIP(hipDeviceGetPCIBusId(out, len, device));
    return AUR_OK;
}

extern "C" int aur_create(int device, aur_ctx** out) {
    AUR_REQUIRE(out != nullptr, "aur_create: null out pointer");
    *out = nullptr;
    int count = 0;
    AUR_CHECK_HIP(hipGetDeviceCount(&count));
    AUR_REQUIRE(device >= 0 && device < count, "aur_create: device %d out of range (have %d)", device, count);
    AUR_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    AUR_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        aur_set_error("aur_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                      prop.gcnArchName);
        return AUR_EUNSUPPORTED;
    }
    aur_ctx* ctx = (aur_ctx*)calloc(1, sizeof(aur_ctx));
    if (!ctx) {
        aur_set_error("aur_create: out of host memory");
        return AUR_ENOMEM;
    }
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount;
    if (const char* e = getenv("AUR_FILL_ROWS")) ctx->fill_rows = atoi(e);
    if (const char* e = getenv("AUR_FILL_NT")) ctx->fill_nt = atoi(e);
    if (const char* e = getenv("AUR_FILL_MODE")) ctx->fill_mode = atoi(e);
    ctx->stft_zt = 1;
    if (const char* e = getenv("AUR_STFT_ZT")) ctx->stft_zt = atoi(e);
    if (const char* e = getenv("AUR_STFT_FPW")) ctx->stft_fpw = atoi(e);
    double h_tw[AUR_TW_DOUBLES];
    build_tables(ctx->h_window, h_tw);
    hipError_t e = hipMalloc((void**)&ctx->d_window, sizeof(float) * AUR_FFT_SIZE);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_twid, sizeof(h_tw));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_rows, (size_t)AUR_ROWS_BYTES + (size_t)AUR_ROWS_FLAGS * 4);
    if (e == hipSuccess)
        e = hipMemcpy(ctx->d_window, ctx->h_window, sizeof(float) * AUR_FFT_SIZE, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_twid, h_tw, sizeof(h_tw), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        aur_set_error("aur_create: table upload failed: %s", hipGetErrorString(e));
        if (ctx->d_window) (void)hipFree(ctx->d_window);
        if (ctx->d_twid) (void)hipFree(ctx->d_twid);
        if (ctx->d_rows) (void)hipFree(ctx->d_rows);
        free(ctx);
        return AUR_EHIP;
    }
    if (int rc = aur_raster_static_init(ctx)) {
        (void)hipFree(ctx->d_window);
        (void)hipFree(ctx->d_twid);
        (void)hipFree(ctx->d_rows);
        free(ctx);
        return rc;
    }
    *out = ctx;
    return AUR_OK;
}

extern "C" void aur_destroy(aur_ctx* ctx) {
    if (!ctx) return;
    if (hipSetDevice(ctx->device) == hipSuccess) {
        if (ctx->d_window) (void)hipFree(ctx->d_window);
        if (ctx->d_twid) (void)hipFree(ctx->d_twid);
        if (ctx->d_rows) (void)hipFree(ctx->d_rows);
    }
    free(ctx);
}

extern "C" int aur_set_background(aur_ctx* ctx, const uint8_t* d_base, int W, int H) {
    AUR_REQUIRE(ctx != nullptr, "aur_set_background: null context");
    if (d_base == nullptr) {
        ctx->d_base = nullptr;
        ctx->base_w = ctx->base_h = 0;
        return AUR_OK;
    }
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_set_background: frame size %dx%d unsupported", W, H);
    AUR_REQUIRE(((uintptr_t)d_base & 15u) == 0, "aur_set_background: the base frame must be 16-byte aligned");
    ctx->d_base = d_base;
    ctx->base_w = W;
    ctx->base_h = H;
    return AUR_OK;
}

int aur_check_background(aur_ctx* ctx, const char* who, int W, int H) {
    AUR_REQUIRE(ctx->d_base == nullptr || (ctx->base_w == W && ctx->base_h == H),
                "%s: the background was set for %dx%d frames, not %dx%d", who, ctx->base_w, ctx->base_h, W, H);
    return AUR_OK;
}

extern "C" int aur_get_window(aur_ctx* ctx, float* h_window) {
    AUR_REQUIRE(ctx && h_window, "aur_get_window: null argument");
    memcpy(h_window, ctx->h_window, sizeof(float) * AUR_FFT_SIZE);
    return AUR_OK;
}

extern "C" int aur_host_alloc(void** out, size_t bytes) {
    AUR_REQUIRE(out != nullptr && bytes > 0, "aur_host_alloc: bad argument");
    AUR_CHECK_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return AUR_OK;
}

extern "C" int aur_host_free(void* p) {
    if (!p) return AUR_OK;
    AUR_CHECK_HIP(hipHostFree(p));
    return AUR_OK;
}

extern "C" int aur_readback_async(aur_ctx* ctx, const void* d_src, size_t bytes, void* h_dst, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(d_src && h_dst, "aur_readback_async: null pointer");
    if (bytes == 0) return AUR_OK;
    AUR_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return AUR_OK;
}

extern "C" int aur_stream_synchronize(aur_ctx* ctx, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return AUR_OK;
}
