// Smoothing scan over frames (gfx950): the frame-to-frame recurrences of the export path, evaluated in the
// reference's sequential float64 order so results are bit-identical to the Python loop.
//
//   energy      widgets/rt_widget.py:824        float(np.clip(np.mean(np.abs(samples)) * sensitivity, 0, 1))
//   _amp_ema    widgets/rt_widget.py:487-499    EMA -> gradient colour with int() truncation
//   flux_smooth audio/analysis.py:29-30         EMA of sum(diff)/1025
//   _phase      widgets/rt_widget.py:825        (phase + 0.04*(0.5+energy)) % (2*pi)
//
// One workgroup (256 threads) per call: the element-wise parts run on all lanes through LDS tiles, the three
// recurrences run on one lane each of three waves at the same time (each is a 2-operation dependent chain per frame).
#include "common.h"

namespace {

constexpr int kTile = 2048;   // frames per LDS tile
constexpr double kTwoPi = 6.283185307179586;   // 2 * np.pi

struct ScanArgs {
    aur_scan_params p;
    const float* meanabs;
    const float* flux_sum;
    int n;
    double* energy;
    uint8_t* pen;
    double* flux_smooth;
    double* phase;
    double* state_out;
};

__global__ __launch_bounds__(256) void scan_frames_kernel(ScanArgs a) {
    __shared__ double s_bx[kTile];     // (1 - amp_alpha) * energy        -> amp_ema after the chain
    __shared__ double s_bf[kTile];     // (1 - flux_alpha) * flux          -> flux_smooth after the chain
    __shared__ double s_ph[kTile];     // 0.04 * (0.5 + energy)            -> phase after the chain
    __shared__ double s_carry[3];

    const int tid = threadIdx.x;
    const double amp_a = a.p.amp_alpha, amp_b = 1.0 - a.p.amp_alpha;
    const double fl_a = a.p.flux_alpha, fl_b = 1.0 - a.p.flux_alpha;
    const float ws = (float)fmax(0.05, a.p.sensitivity);     // np.float32 * python float -> float32 (NEP 50)
    if (tid == 0) {
        s_carry[0] = a.p.amp_ema0;
        s_carry[1] = a.p.flux_smooth0;
        s_carry[2] = a.p.phase0;
    }
    __syncthreads();

    for (int base = 0; base < a.n; base += kTile) {
        const int m = (a.n - base < kTile) ? a.n - base : kTile;
        // element-wise front end
        for (int i = tid; i < m; i += 256) {
            float e32 = a.meanabs[base + i] * ws;
            e32 = fminf(fmaxf(e32, 0.0f), 1.0f);
            const double e = (double)e32;
            if (a.energy) a.energy[base + i] = e;
            s_bx[i] = amp_b * e;
            s_bf[i] = fl_b * ((double)a.flux_sum[base + i] / 1025.0);
            s_ph[i] = 0.04 * (0.5 + e);
        }
        __syncthreads();
        // the three sequential chains, one lane each; LDS reads run U elements ahead of the dependent chain
        // (lane 0 of waves 0, 1, 2: three SIMDs run the three chains side by side instead of one wave serialising
        // its divergent lanes)
        if ((tid & 63) == 0 && tid < 192) {
            constexpr int U = 8;
            const int c = tid >> 6;
            double s = s_carry[c];
            double* arr = (c == 0) ? s_bx : (c == 1) ? s_bf : s_ph;
            const double a = (c == 0) ? amp_a : fl_a;
            int i = 0;
            if (c < 2) {
                for (; i + U <= m; i += U) {
                    double b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) b[u] = arr[i + u];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        s = a * s + b[u];
                        arr[i + u] = s;
                    }
                }
                for (; i < m; ++i) {
                    s = a * s + arr[i];
                    arr[i] = s;
                }
            } else {
                for (; i + U <= m; i += U) {
                    double b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) b[u] = arr[i + u];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        s = s + b[u];
                        if (s >= kTwoPi) s = s - kTwoPi;   // fmod of a value in [0, 2*(2 pi)) is this exact subtraction
                        arr[i + u] = s;
                    }
                }
                for (; i < m; ++i) {
                    s = s + arr[i];
                    if (s >= kTwoPi) s = s - kTwoPi;
                    arr[i] = s;
                }
            }
            s_carry[c] = s;
        }
        __syncthreads();
        // element-wise back end: colour ramp with int() truncation, stores
        for (int i = tid; i < m; i += 256) {
            if (a.pen) {
                double t = (s_bx[i] - a.p.grad_min) / fmax(1e-6, a.p.grad_max - a.p.grad_min);
                t = fmax(0.0, fmin(1.0, t));
                if (a.p.grad_curve == 1) {
                    t = t * t;
                } else if (a.p.grad_curve == 2) {
                    t = 1.0 - (1.0 - t) * (1.0 - t);
                } else if (a.p.grad_curve == 3) {
                    t = t * t * (3.0 - 2.0 * t);
                }
                uint32_t px = 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double ga = (double)a.p.grad_a[c], gb = (double)a.p.grad_b[c];
                    const int v = (int)(ga + t * (gb - ga));
                    px |= ((uint32_t)(v & 0xff)) << (8 * c);
                }
                reinterpret_cast<uint32_t*>(a.pen)[base + i] = px;
            }
            if (a.flux_smooth) a.flux_smooth[base + i] = s_bf[i];
            if (a.phase) a.phase[base + i] = s_ph[i];
        }
        __syncthreads();
    }
    if (a.state_out && tid < 3) a.state_out[tid] = s_carry[tid];
}

}  // namespace

extern "C" int aur_scan_frames(aur_ctx* ctx, const float* d_meanabs, const float* d_flux_sum, int n_frames,
                               const aur_scan_params* params, double* d_energy, uint8_t* d_pen_rgb,
                               double* d_flux_smooth, double* d_phase, double* d_state_out, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(params != nullptr, "aur_scan_frames: null params");
    AUR_REQUIRE(n_frames >= 0, "aur_scan_frames: n_frames %d < 0", n_frames);
    AUR_REQUIRE(n_frames == 0 || (d_meanabs && d_flux_sum), "aur_scan_frames: null input pointer");   // 0 frames: the
    // kernel still copies the carried state to d_state_out
    AUR_REQUIRE(params->grad_curve >= 0 && params->grad_curve <= 3, "aur_scan_frames: grad_curve %d unknown",
                params->grad_curve);
    AUR_REQUIRE(params->amp_alpha >= 0.0 && params->amp_alpha <= 1.0, "aur_scan_frames: amp_alpha out of [0,1]");
    AUR_REQUIRE(d_pen_rgb == nullptr || ((uintptr_t)d_pen_rgb & 3u) == 0, "aur_scan_frames: d_pen_rgb misaligned");
    ScanArgs a;
    a.p = *params;
    a.meanabs = d_meanabs;
    a.flux_sum = d_flux_sum;
    a.n = n_frames;
    a.energy = d_energy;
    a.pen = d_pen_rgb;
    a.flux_smooth = d_flux_smooth;
    a.phase = d_phase;
    a.state_out = d_state_out;
    hipLaunchKernelGGL(scan_frames_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}
