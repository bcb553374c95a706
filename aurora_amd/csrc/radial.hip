// "Spectrum - Radial" for gfx950 (MI355X): the radial spectrum path with its spatial / temporal smoothing, the
// centre-disc ("feather") envelope, bit-plane rasterisation of both, and the RGB24 fill that composes them.
//
// Replaces widgets/rt_widget.py:942-981 (_draw_spectrum, radial branch), :1121-1137 (_smooth_closed),
// :953-958 (_radial_prev temporal EMA), :1004-1051 + :1056-1096 (_draw_center_image_and_mask: envelope path and
// QColor(30,30,40) fill) for the export entry point paint_frame (:821-852).
//
// Arithmetic follows the reference's numpy/Python expression order in float64 with contraction off (the library is
// built with -ffp-contract=off), so vertices are bit-identical to the oracle for the export defaults:
//  * np.interp on an integer grid: slope * (xi - j) + a[j] with slope = a[j+1] - a[j]; xi = i * step (np.linspace);
//  * np.convolve with an all-equal kernel: sequential sum of a[i+k] * (1/win); numpy switches to an FMA dot product
//    for kernels longer than 11 taps, which is mirrored here (win = 13 only occurs at amount = 100);
//  * angles are host tables of (cos, sin) computed with the reference's own expression (index-map rule);
//  * fbm / tanh of the envelope (only active with edge_waviness > 0) use the device libm: a few ulp, not bit-exact.
// Coverage rule: oracle/raster_ref.py (pixel centre inside the closed outline; disc edge not antialiased).
#include "common.h"

namespace {

constexpr double kBig = 1.0e6;
constexpr int kMaxL = 4096;          // longest smoothed ring (158 * 5 * 5 = 3950)

// ---- _smooth_closed (rt_widget.py:1121-1137) on an LDS array ------------------------------------------------------
// in[L] -> out[L * up]; tmp[L * up] is scratch.  Whole block cooperates; ends with a barrier.
__device__ void smooth_closed_lds(const double* in, int L, int amount, double step, double* tmp, double* out) {
    const int up = 1 + amount / 25;
    const int n = L * up;
    const int half = (int)((double)amount * 0.06);
    const int win = 2 * half + 1;
    double* a = (win > 1) ? tmp : out;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double v;
        if (up > 1) {
            if (i == n - 1) {
                v = in[L - 1];                                   // linspace endpoint == x[L-1]
            } else {
                const double xi = (double)i * step;
                int j = (int)xi;
                if (j > L - 2) j = L - 2;
                const double xj = (double)j;
                if (xj == xi) {
                    v = in[j];
                } else {
                    const double slope = in[j + 1] - in[j];      // / (x[j+1] - x[j]) == 1
                    v = slope * (xi - xj) + in[j];
                }
            }
        } else {
            v = in[i];
        }
        a[i] = v;
    }
    __syncthreads();
    if (win > 1) {
        const double c = 1.0 / (double)win;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            double s = 0.0;
            if (win >= 12) {
                for (int k = 0; k < win; ++k) {
                    int idx = i - half + k;
                    idx = (idx % n + n) % n;
                    s = fma(a[idx], c, s);
                }
            } else {
                for (int k = 0; k < win; ++k) {
                    int idx = i - half + k;
                    idx = (idx % n + n) % n;
                    s = s + a[idx] * c;
                }
            }
            out[i] = s;
        }
        __syncthreads();
    }
}

// spectrum[:80] / max -> (mirror) -> _smooth_closed(radial_wave_smoothness) -> vals1[f][L1]   (:938-951)
__global__ __launch_bounds__(256) void radial_stage1_kernel(const float* __restrict__ logspec, int mirror,
                                                            int amount1, double step1, int L0, int L1,
                                                            double* __restrict__ vals1) {
    __shared__ double s_a[160];
    __shared__ double s_t[kMaxL / 4];
    __shared__ double s_o[kMaxL / 4];
    __shared__ float s_red[2];
    const int f = blockIdx.x;
    const float* spec = logspec + (long)f * AUR_NBINS;
    float v = (threadIdx.x < AUR_N_BARS) ? spec[threadIdx.x] : -INFINITY;
    if (threadIdx.x < 128) {
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    }
    __syncthreads();
    const float mxv = fmaxf(s_red[0], s_red[1]);
    const float mx = (mxv > 0.0f) ? mxv : 1.0f;                  // rt_widget.py:939
    if (threadIdx.x < AUR_N_BARS) {
        const double n = (double)__fdiv_rn(spec[threadIdx.x], mx);   // float32 divide (:940), widened by _smooth_closed
        s_a[threadIdx.x] = n;
        // np.concatenate([spec, spec[::-1][1:-1]]): element 80 + m = spec[78 - m], m = 0..77   (:946)
        if (mirror && threadIdx.x >= 1 && threadIdx.x <= AUR_N_BARS - 2) s_a[2 * AUR_N_BARS - 2 - threadIdx.x] = n;
    }
    __syncthreads();
    double* dst = vals1 + (long)f * L1;
    if (L0 > 3 && amount1 > 0) {
        smooth_closed_lds(s_a, L0, amount1, step1, s_t, s_o);
        for (int i = threadIdx.x; i < L1; i += blockDim.x) dst[i] = s_o[i];
    } else {
        for (int i = threadIdx.x; i < L1; i += blockDim.x) dst[i] = s_a[i];
    }
}

// _radial_prev = alpha * _radial_prev + (1 - alpha) * spec_draw over consecutive frames, in place (:953-958).
// One thread per ring element; the first frame of a track copies (prev is None), a shard continues from `state`.
__global__ __launch_bounds__(256) void radial_ema_kernel(double* __restrict__ vals1, int n_frames, int L1,
                                                         double alpha, int has_state, double* __restrict__ state) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L1) return;
    const double beta = 1.0 - alpha;
    double prev = has_state ? state[i] : 0.0;
    int f = 0;
    if (!has_state && n_frames > 0) {
        prev = vals1[i];
        f = 1;
    }
    constexpr int U = 16;                     // loads of U frames in flight per thread; the chain itself is serial
    for (; f + U <= n_frames; f += U) {
        double cur[U];
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = vals1[(long)(f + u) * L1 + i];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            prev = alpha * prev + beta * cur[u];
            vals1[(long)(f + u) * L1 + i] = prev;
        }
    }
    for (; f < n_frames; ++f) {
        const double cur = vals1[(long)f * L1 + i];
        prev = alpha * prev + beta * cur;
        vals1[(long)f * L1 + i] = prev;
    }
    state[i] = prev;
}

// _smooth_closed(radial_smooth_amount) -> r = inner + span * v -> (cx + cos a * r, cy + sin a * r)   (:960-974)
__global__ __launch_bounds__(256) void radial_vertices_kernel(const double* __restrict__ vals1,
                                                              const double* __restrict__ energy, int L1, int amount2,
                                                              double step2, int L2, const double* __restrict__ cs,
                                                              int W, int H, double* __restrict__ verts) {
    extern __shared__ double s_dyn[];          // [L1] input ring, [L2] scratch, [L2] smoothed ring
    double* s_a = s_dyn;
    double* s_t = s_dyn + L1;
    double* s_o = s_t + L2;
    const int f = blockIdx.x;
    const double* src = vals1 + (long)f * L1;
    for (int i = threadIdx.x; i < L1; i += blockDim.x) s_a[i] = src[i];
    __syncthreads();
    const double* vals = s_a;
    if (L1 > 3 && amount2 > 0) {
        smooth_closed_lds(s_a, L1, amount2, step2, s_t, s_o);
        vals = s_o;
    }
    const double mn = (double)(W < H ? W : H);
    const double cx = (double)W / 2.0, cy = (double)H / 2.0;
    const double inner = mn * 0.22;                               // :944
    const double span = mn * 0.28 * (0.6 + 0.6 * energy[f]);      // :945
    double* dst = verts + (long)f * (L2 + 1) * 2;
    for (int i = threadIdx.x; i <= L2; i += blockDim.x) {
        const double v = vals[i < L2 ? i : L2 - 1];               // spec_draw[min(i, L - 1)]
        const double r = inner + span * v;
        dst[2 * i] = cx + cs[2 * i] * r;
        dst[2 * i + 1] = cy + cs[2 * i + 1] * r;
    }
}

struct EnvArgs {
    int mirror, W, H;
    double noise_amt;      // edge_waviness / 100
    double audio_amt;      // feather_audio_amount / 100 if enabled else 0
    double sens_ratio;     // feather_sensitivity / max(1e-3, waveform_sensitivity)
    double rot;            // np.deg2rad(radial_rotation_deg)
};

// Envelope of the centre disc: 361 vertices per frame (:1007-1051).
__global__ __launch_bounds__(384) void envelope_vertices_kernel(const float* __restrict__ logspec,
                                                                const double* __restrict__ energy,
                                                                const double* __restrict__ phase, EnvArgs a,
                                                                const double* __restrict__ cs360,
                                                                double* __restrict__ verts) {
    __shared__ float s_red[8];
    const int f = blockIdx.x;
    const float* spec = logspec + (long)f * AUR_NBINS;
    const double mn = (double)(a.W < a.H ? a.W : a.H);
    const double cx = (double)a.W / 2.0, cy = (double)a.H / 2.0;
    const double inner = mn * 0.22;
    float mxv = 0.0f;
    if (a.audio_amt != 0.0) {     // max over the whole spectrum (:1024); skipped when the audio term is off
        float m = -INFINITY;
        for (int i = threadIdx.x; i < AUR_NBINS; i += blockDim.x) m = fmaxf(m, spec[i]);
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
        __syncthreads();
        mxv = s_red[0];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) mxv = fmaxf(mxv, s_red[k]);
    }
    const float mx = (mxv > 1e-9f) ? mxv : 1.0f;
    const double e = energy[f];
    const double amp_noise_px = mn * 0.025 * a.noise_amt;                               // :1029
    const double energy_feather = fmin(1.0, e * a.sens_ratio);                          // :1030
    const double amp_audio_px = mn * 0.06 * a.audio_amt * (0.4 + 0.6 * energy_feather);  // :1031
    const int i = threadIdx.x;
    if (i > 360) return;
    const int nlen = a.mirror ? 2 * AUR_NBINS - 2 : AUR_NBINS;
    double s_val = 0.0;
    if (a.audio_amt != 0.0) {
        int si = (int)(((double)i / 360.0) * (double)(nlen - 1));                       // :1036
        if (si >= AUR_NBINS) si = 2 * AUR_NBINS - 2 - si;      // mirrored tail: element 1025 + m = base[1023 - m]
        s_val = (double)__fdiv_rn(spec[si], mx);
    }
    double fb = 0.0;
    if (amp_noise_px != 0.0) {
        const double ang = 2.0 * 3.141592653589793 * ((double)i / 360.0) - (3.141592653589793 / 2.0) + a.rot;
        const double theta = ang * 2.0, ph = phase[f];
        double v = 0.0, amp = 1.0, freq = 1.0;
        for (int k = 0; k < 3; ++k) {                                                   // fbm, :1010-1018
            v += amp * sin(freq * theta + (double)(k + 1) * ph);
            amp *= 0.5;
            freq *= 2.0;
        }
        fb = v / 1.75;
    }
    double r = inner + amp_noise_px * fb + amp_audio_px * s_val;                        // :1038
    const double r_max = 0.48 * mn;
    if (r_max > inner) {                                                                // :1041-1044
        double g = fmax(0.0, (r - inner) / (r_max - inner));
        if (g != 0.0) g = tanh(1.25 * g) / 0.84828363995751289761;    // math.tanh(1.25)
        r = inner + g * (r_max - inner);
    }
    double* dst = verts + ((long)f * 361 + i) * 2;
    dst[0] = cx + cs360[2 * i] * r;
    dst[1] = cy + cs360[2 * i + 1] * r;
}

// Feather rings (:1098-1114): six scaled copies k = 1..6 of the path
//   r = (inner + amp_noise sin(2 a + phase) + amp_audio spec_norm[si]) * (1 + 0.015 k)
// (a plain sine instead of the fbm, no tanh limit).  Block = one frame, thread = one vertex; rings [F, 6, 361, 2].
__global__ __launch_bounds__(384) void feather_ring_vertices_kernel(const float* __restrict__ logspec,
                                                                    const double* __restrict__ energy,
                                                                    const double* __restrict__ phase, EnvArgs a,
                                                                    const double* __restrict__ cs360,
                                                                    double* __restrict__ rings) {
    __shared__ float s_red[8];
    const int f = blockIdx.x;
    const float* spec = logspec + (long)f * AUR_NBINS;
    const double mn = (double)(a.W < a.H ? a.W : a.H);
    const double cx = (double)a.W / 2.0, cy = (double)a.H / 2.0;
    const double inner = mn * 0.22;
    float mxv = 0.0f;
    if (a.audio_amt != 0.0) {
        float m = -INFINITY;
        for (int i = threadIdx.x; i < AUR_NBINS; i += blockDim.x) m = fmaxf(m, spec[i]);
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
        __syncthreads();
        mxv = s_red[0];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) mxv = fmaxf(mxv, s_red[k]);
    }
    const float mx = (mxv > 1e-9f) ? mxv : 1.0f;
    const double amp_noise_px = mn * 0.025 * a.noise_amt;
    const double energy_feather = fmin(1.0, energy[f] * a.sens_ratio);
    const double amp_audio_px = mn * 0.06 * a.audio_amt * (0.4 + 0.6 * energy_feather);
    const int i = threadIdx.x;
    if (i > 360) return;
    const int nlen = a.mirror ? 2 * AUR_NBINS - 2 : AUR_NBINS;
    double s_val = 0.0;
    if (a.audio_amt != 0.0) {
        int si = (int)(((double)i / 360.0) * (double)(nlen - 1));
        if (si >= AUR_NBINS) si = 2 * AUR_NBINS - 2 - si;
        s_val = (double)__fdiv_rn(spec[si], mx);
    }
    double sn = 0.0;
    if (amp_noise_px != 0.0) {
        const double ang = 2.0 * 3.141592653589793 * ((double)i / 360.0) - (3.141592653589793 / 2.0) + a.rot;
        sn = sin(ang * 2.0 + phase[f]);
    }
    const double r0 = inner + amp_noise_px * sn + amp_audio_px * s_val;                 // :1107
    for (int k = 1; k <= 6; ++k) {
        const double r = r0 * (1.0 + 0.015 * (double)k);
        double* dst = rings + (((long)f * 6 + (k - 1)) * 361 + i) * 2;
        dst[0] = cx + cs360[2 * i] * r;
        dst[1] = cy + cs360[2 * i + 1] * r;
    }
}

// ---- "Waveform - Circular" -----------------------------------------------------------------------------------------
// rt_widget.py:871-896 with n = 1024 samples: segs = min(720, n), step = max(1, int(n / segs)) = 1, so every sample is
// a vertex: r = base_r + float(samples[i]) * amp, (x, y) = (cx + cos(a_i) r, cy + sin(a_i) r).  One thread per vertex.
__global__ __launch_bounds__(256) void wave_circular_vertices_kernel(const float* __restrict__ samples,
                                                                     const double* __restrict__ energy,
                                                                     const double* __restrict__ cs, int W, int H,
                                                                     double* __restrict__ verts) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;          // < AUR_BLOCK
    const double mn = (double)(W < H ? W : H);
    const double cx = (double)W / 2.0, cy = (double)H / 2.0;
    const double base_r = mn * 0.28;                                        // :873
    const double amp = mn * 0.22 * (0.5 + 0.6 * energy[f]);                 // :874
    const double v = (double)samples[(long)f * AUR_BLOCK + i];              // :886
    const double r = base_r + v * amp;                                      // :888
    double* dst = verts + ((long)f * AUR_BLOCK + i) * 2;
    dst[0] = cx + cs[2 * i] * r;                                            // :889
    dst[1] = cy + cs[2 * i + 1] * r;                                        // :890
}

// ---- "Waveform - Linear" as a path (only needed when its shadow is drawn) ---------------------------------------------
// rt_widget.py:904-913: vertex x is (x, float(mid - samples[i] * amp)), i = col2samp[x]; the product and the difference
// are float32 (np.float32 with Python floats, NEP 50), exactly as in spans_waveform_kernel (raster.hip).
__global__ __launch_bounds__(256) void wave_linear_vertices_kernel(const float* __restrict__ samples,
                                                                   const int* __restrict__ col2samp,
                                                                   const double* __restrict__ energy, int W, int H,
                                                                   double* __restrict__ verts) {
    const int f = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const double amp = 0.45 * ((double)H / 2.0) * (0.5 + energy[f]);
    const float amp32 = (float)amp;
    const float mid32 = (float)((double)H / 2.0);
    double* dst = verts + ((long)f * W + x) * 2;
    dst[0] = (double)x;
    dst[1] = (double)(mid32 - samples[(long)f * AUR_BLOCK + col2samp[x]] * amp32);
}

// ---- bit planes -------------------------------------------------------------------------------------------------

__device__ __forceinline__ void or_span(uint32_t* row, int c0, int c1, int W) {
    if (c0 < 0) c0 = 0;
    if (c1 > W) c1 = W;
    if (c1 <= c0) return;
    const int w0 = c0 >> 5, w1 = (c1 - 1) >> 5;
    for (int w = w0; w <= w1; ++w) {
        const int lo = (w == w0) ? (c0 & 31) : 0;
        const int hi = (w == w1) ? ((c1 - 1) & 31) : 31;
        const uint32_t bits = (0xffffffffu >> (31 - hi)) & (0xffffffffu << lo);
        atomicOr(row + w, bits);
    }
}

// Where the pixel runs of a stroke go: into a bit plane (atomicOr), or -- PaintSink further down -- straight into a frame.
struct PlaneSink {
    uint32_t* plane;
    int W32, W;
    __device__ __forceinline__ void span(long r, int c0, int c1) const { or_span(plane + r * W32, c0, c1, W); }
};

// Rows of a convex polygon (n <= 4 vertices) handed to the sink as pixel runs.
// Rows r0 + lane_k, r0 + lane_k + lane_n, .. only: several lanes can share one primitive (the frame painter does).
template <class Sink>
__device__ void raster_convex(const Sink& sink, const double* px, const double* py, int n, int H, int lane_k = 0,
                              int lane_n = 1) {
    double ymin = py[0], ymax = py[0];
    for (int i = 1; i < n; ++i) {
        ymin = fmin(ymin, py[i]);
        ymax = fmax(ymax, py[i]);
    }
    if (!(ymin == ymin) || !(ymax == ymax)) return;      // NaN vertices: nothing drawn
    ymin = fmin(fmax(ymin, -kBig), kBig);
    ymax = fmin(fmax(ymax, -kBig), kBig);
    long r0 = (long)ceil(ymin - 0.5), r1 = (long)floor(ymax - 0.5) + 1;
    if (r0 < 0) r0 = 0;
    if (r1 > H) r1 = H;
    for (long r = r0 + lane_k; r < r1; r += lane_n) {
        const double yc = (double)r + 0.5;
        double xmin = kBig, xmax = -kBig;
        bool hit = false;
        for (int e = 0; e < n; ++e) {
            const int e1 = (e + 1 == n) ? 0 : e + 1;
            const double x0 = px[e], y0 = py[e], x1 = px[e1], y1 = py[e1];
            if (y0 == y1) {
                if (y0 == yc) {
                    xmin = fmin(xmin, fmin(x0, x1));
                    xmax = fmax(xmax, fmax(x0, x1));
                    hit = true;
                }
            } else if ((yc - y0) * (yc - y1) <= 0.0) {
                const double x = x0 + ((yc - y0) * (x1 - x0)) / (y1 - y0);
                xmin = fmin(xmin, x);
                xmax = fmax(xmax, x);
                hit = true;
            }
        }
        if (!hit) continue;
        xmin = fmin(fmax(xmin, -kBig), kBig);
        xmax = fmin(fmax(xmax, -kBig), kBig);
        sink.span(r, (int)ceil(xmin - 0.5), (int)floor(xmax - 0.5) + 1);
    }
}

__device__ __forceinline__ bool seg_dir(const double* V, int n, int s, double& ux, double& uy) {
    const int s1 = (s + 1 == n) ? 0 : s + 1;
    const double dx = V[2 * s1] - V[2 * s], dy = V[2 * s1 + 1] - V[2 * s + 1];
    const double ln = sqrt(dx * dx + dy * dy);
    if (ln == 0.0) return false;
    ux = dx / ln;
    uy = dy / ln;
    return true;
}

// BevelJoin stroke of half-width hw (1 for the width-2 pen) of the closed polygon verts[f][nv][2] translated by (tx, ty):
// thread t < nv draws segment t, thread nv + t the bevel at vertex t.  The plane must be zero on entry.  hw = 1 and
// tx = ty = 0 reproduce the untranslated width-2 stroke bit for bit (x + 0 and n * 1 are exact).
template <class Sink>
__device__ __forceinline__ void stroke_primitive(const double* __restrict__ V, int nv, int t, int H, double hw, double tx,
                                                 double ty, int closed, int round_join, const Sink& sink, int lane_k = 0,
                                                 int lane_n = 1) {
    // closed == 0: open polyline (the envelope path of `_draw_center_image_and_mask` has no closeSubpath): no segment
    // from the last vertex to the first, no join at either end, SquareCap = the end segments extended by hw.
    if (t >= 2 * nv) return;
    // round_join != 0 (RoundCap / RoundJoin pen): plain segment rectangles and a disc of radius hw at every vertex,
    // i.e. every pixel centre within hw of the polyline.
    if (!closed && t == nv - 1) return;
    if (!closed && !round_join && (t == nv || t == 2 * nv - 1)) return;
    double px[4], py[4];
    if (t < nv) {
        double ux, uy;
        if (!seg_dir(V, nv, t, ux, uy)) return;
        const int t1 = (t + 1 == nv) ? 0 : t + 1;
        double ax = V[2 * t] + tx, ay = V[2 * t + 1] + ty, bx = V[2 * t1] + tx, by = V[2 * t1 + 1] + ty;
        if (!closed && !round_join && t == 0) {
            ax = ax - ux * hw;
            ay = ay - uy * hw;
        }
        if (!closed && !round_join && t == nv - 2) {
            bx = bx + ux * hw;
            by = by + uy * hw;
        }
        const double nx = -uy * hw, ny = ux * hw;
        px[0] = ax + nx; py[0] = ay + ny;
        px[1] = bx + nx; py[1] = by + ny;
        px[2] = bx - nx; py[2] = by - ny;
        px[3] = ax - nx; py[3] = ay - ny;
        raster_convex(sink, px, py, 4, H, lane_k, lane_n);
    } else {
        const int v = t - nv;
        if (round_join) {
            const double vx = V[2 * v] + tx, vy = V[2 * v + 1] + ty;
            if (!(vx == vx) || !(vy == vy)) return;
            const double yl = fmin(fmax(vy - hw, -kBig), kBig), yh = fmin(fmax(vy + hw, -kBig), kBig);
            long r0 = (long)ceil(yl - 0.5), r1 = (long)floor(yh - 0.5) + 1;
            if (r0 < 0) r0 = 0;
            if (r1 > H) r1 = H;
            for (long r = r0 + lane_k; r < r1; r += lane_n) {
                const double dyc = ((double)r + 0.5) - vy;
                const double c2 = hw * hw - dyc * dyc;
                if (c2 < 0.0) continue;
                const double c = sqrt(c2);
                const double xl = fmin(fmax(vx - c, -kBig), kBig), xh = fmin(fmax(vx + c, -kBig), kBig);
                sink.span(r, (int)ceil(xl - 0.5), (int)floor(xh - 0.5) + 1);
            }
            return;
        }
        const int a = (v == 0) ? nv - 1 : v - 1;
        double uxa, uya, uxb, uyb;
        if (!seg_dir(V, nv, a, uxa, uya) || !seg_dir(V, nv, v, uxb, uyb)) return;
        const double cr = uxa * uyb - uya * uxb;
        if (cr == 0.0) return;
        const double sg = (cr > 0.0) ? -1.0 : 1.0;
        const double vx = V[2 * v] + tx, vy = V[2 * v + 1] + ty;
        px[0] = vx; py[0] = vy;
        px[1] = vx + sg * (-uya) * hw; py[1] = vy + sg * uxa * hw;
        px[2] = vx + sg * (-uyb) * hw; py[2] = vy + sg * uxb * hw;
        raster_convex(sink, px, py, 3, H, lane_k, lane_n);
    }
}

__global__ __launch_bounds__(256) void mask_stroke_closed_kernel(const double* __restrict__ verts, int nv, int W, int H,
                                                                 int W32, double hw, double tx, double ty,
                                                                 int closed, int round_join,
                                                                 uint32_t* __restrict__ planes) {
    const int f = blockIdx.y;
    const PlaneSink sink = {planes + (long)f * H * W32, W32, W};
    stroke_primitive(verts + (long)f * nv * 2, nv, blockIdx.x * 256 + threadIdx.x, H, hw, tx, ty, closed, round_join, sink);
}

// Even-odd fill of the implicitly closed polygon verts[f][nv][2], sampled at pixel centres.  A pixel is inside iff an odd
// number of edge crossings of its row lie at or left of it, so every crossing simply TOGGLES one bit of the row -- the
// first pixel at or right of it -- in an LDS bitmap (atomicXor: order-free, any number of crossings per row, nothing to
// sort or to overflow), and the fill is the prefix XOR of that bitmap along the row.  Rows with an odd number of
// crossings (only NaN vertices produce them) drop their right-most crossing, as the sorted-pair rule of the oracle does.
// The block then writes its rows as one contiguous piece of the plane; every word is written, so the plane needs no
// clearing.
__device__ __forceinline__ int span_edge(double x, int bx0, int W) {     // column relative to bx0, clamped to [0, W]
    const double a = fmin(fmax(x, -kBig), kBig);
    const long c = (long)ceil(a - 0.5) - bx0;
    return c < 0 ? 0 : (c > W ? W : (int)c);
}
constexpr int kPolyWords = 8192;     // LDS bitmap of a block: rows_per_block * W32 <= this (32 KB)
// The plane may cover only a box of the frame (aur_mask_polygon_box): pixel columns [bx0, bx0 + W), rows [by0, by0 + H),
// bx0 a multiple of 32; W, H, W32 are then the box's.  Crossings are computed in frame coordinates -- the arithmetic of the
// full-frame plane, bit for bit -- and only the resulting pixel column and row are moved into the box.
__global__ __launch_bounds__(128) void mask_polygon_kernel(const double* __restrict__ verts, int nv, int W, int H,
                                                           int W32, int bx0, int by0, int rows_per_block, int vec_ok,
                                                           uint32_t* __restrict__ planes) {
    __shared__ uint32_t s_bits[kPolyWords];      // [row][word]: toggles, then the filled words
    __shared__ uint8_t s_par[kPolyWords];        // parity of the toggles of a word, then the carry into it
    __shared__ int s_cnt[128], s_max[128];       // crossings of a row and the right-most toggle position
    const int f = blockIdx.y;
    const double* V = verts + (long)f * nv * 2;      // edges read their end points from global memory (L2-resident)
    const int base = blockIdx.x * rows_per_block;
    const int rows_here = (H - base < rows_per_block) ? H - base : rows_per_block;
    const int total = rows_here * W32;
    for (int idx = threadIdx.x; idx < total; idx += 128) s_bits[idx] = 0u;
    if (threadIdx.x < rows_per_block) {
        s_cnt[threadIdx.x] = 0;
        s_max[threadIdx.x] = -1;
    }
    int found = 0;
    __syncthreads();
    // Edge-parallel crossing search: an edge crosses the sample line of row r (y = r + 0.5) iff exactly one of its end
    // points is <= it, i.e. for the rows between ceil(ymin - 0.5) and ceil(ymax - 0.5) - 1; a thread walks the few rows
    // of its edges (one candidate row more on either side, every row decided by the exact predicate, so rounding of
    // the bounds cannot change the result).
    for (int e = threadIdx.x; e < nv; e += 128) {
        const int ep = (e == 0) ? nv - 1 : e - 1;
        const double x0 = V[2 * ep], y0 = V[2 * ep + 1], x1 = V[2 * e], y1 = V[2 * e + 1];
        double ya = fmin(y0, y1), yb = fmax(y0, y1);
        if (!(ya == ya) || !(yb == yb)) continue;                    // NaN vertex: no crossing (comparisons are false)
        ya = fmin(fmax(ya, -kBig), kBig);
        yb = fmin(fmax(yb, -kBig), kBig);
        long ra = (long)ceil(ya - 0.5) - 1, rb = (long)ceil(yb - 0.5);
        if (ra < (long)by0 + base) ra = (long)by0 + base;
        if (rb > (long)by0 + base + rows_here - 1) rb = (long)by0 + base + rows_here - 1;
        for (long r = ra; r <= rb; ++r) {
            const double yc = (double)r + 0.5;
            if ((y0 <= yc) != (y1 <= yc)) {
                const double x = x0 + ((yc - y0) * (x1 - x0)) / (y1 - y0);
                const int c = span_edge(x, bx0, W);                  // first pixel whose centre is at or right of x
                const int rr = (int)(r - by0 - base);
                atomicAdd(&s_cnt[rr], 1);
                atomicMax(&s_max[rr], c);
                if (c < W) atomicXor(&s_bits[rr * W32 + (c >> 5)], 1u << (c & 31));
                found = 1;
            }
        }
    }
    uint32_t* dst = planes + ((long)f * H + base) * W32;
    if (__syncthreads_or(found) == 0) {                               // no edge crosses these rows: zeros
        if (vec_ok) {                                                 // W32 % 4 == 0 and a 16-byte aligned plane
            uint4* d4 = reinterpret_cast<uint4*>(dst);
            for (int idx = threadIdx.x; idx < total / 4; idx += 128) d4[idx] = uint4{0u, 0u, 0u, 0u};
        } else {
            for (int idx = threadIdx.x; idx < total; idx += 128) dst[idx] = 0u;
        }
        return;
    }
    if (threadIdx.x < rows_here && (s_cnt[threadIdx.x] & 1)) {        // odd count: the right-most crossing is dropped
        const int c = s_max[threadIdx.x];
        if (c >= 0 && c < W) s_bits[threadIdx.x * W32 + (c >> 5)] ^= 1u << (c & 31);
    }
    __syncthreads();
    // prefix XOR inside every word, parity of every word
    for (int idx = threadIdx.x; idx < total; idx += 128) {
        uint32_t t = s_bits[idx];
        s_par[idx] = (uint8_t)(__builtin_popcount(t) & 1);
        t ^= t << 1;
        t ^= t << 2;
        t ^= t << 4;
        t ^= t << 8;
        t ^= t << 16;
        s_bits[idx] = t;
    }
    __syncthreads();
    if (threadIdx.x < rows_here) {                                    // carry along the row: inside / outside at each word start
        uint8_t* par = s_par + threadIdx.x * W32;
        uint8_t carry = 0;
        for (int w = 0; w < W32; ++w) {
            const uint8_t p = par[w];
            par[w] = carry;
            carry ^= p;
        }
    }
    __syncthreads();
    const uint32_t tail = (W & 31) ? ((1u << (W & 31)) - 1u) : 0xffffffffu;   // bits past the last pixel stay 0
    const uint32_t inv = 0xffffffffu / (uint32_t)W32 + 1u;            // idx / W32 == (idx * inv) >> 32 for idx < 2^16
    for (int idx = threadIdx.x; idx < total; idx += 128) {
        const int rr = (int)__umulhi((uint32_t)idx, inv);
        const int w = idx - rr * W32;
        uint32_t bits = s_bits[idx] ^ (s_par[idx] ? 0xffffffffu : 0u);
        if (w == W32 - 1) bits &= tail;
        dst[idx] = bits;
    }
}

// ---- fill: two bit planes -> RGB24 (same 16-byte-column layout as raster_fill16_kernel) -----------------------------

constexpr uint32_t kBg24 = 13u | (13u << 8) | (18u << 16);     // rt_widget.py:829
constexpr uint32_t kDisc24 = 30u | (30u << 8) | (40u << 16);   // rt_widget.py:1092/1094
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

// One wave owns 64 uint4 columns (up to 12 mask words per row and plane) x 16 rows.  It first pulls the 2 x 16 x 12
// mask words of its tile with six coalesced loads per lane; a tile without any set bit (most of a frame) is pure
// background stores, otherwise the words are parked in LDS and every row extracts its 6 pixel bits from there.
constexpr int kTileWords = 12;
__global__ __launch_bounds__(256) void raster_fill_planes_kernel(const uint32_t* __restrict__ stroke,
                                                                 const uint32_t* __restrict__ disc,
                                                                 const uint32_t* __restrict__ pen_rgb, int W, int H,
                                                                 int W32, int nu4, int nchunks, int disc_static,
                                                                 uint4* __restrict__ frames) {
    __shared__ uint32_t s_m[4][2][kRowsPerTask][kTileWords];
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wave;
    const int chunk = task % nchunks, tile = task / nchunks;
    const int row0 = tile * kRowsPerTask;
    if (row0 >= H) return;                                   // wave-uniform
    const int row1 = (row0 + kRowsPerTask < H) ? row0 + kRowsPerTask : H;
    const int wbase = ((1024 * chunk) / 3) >> 5;             // first mask word of the chunk's pixel range
    uint32_t any = 0;
    uint32_t* sm = &s_m[wave][0][0][0];
#pragma unroll
    for (int k = 0; k < 2 * kRowsPerTask * kTileWords / 64; ++k) {
        const int idx = lane + 64 * k;
        const int plane = idx / (kRowsPerTask * kTileWords), rem = idx % (kRowsPerTask * kTileWords);
        const int r = rem / kTileWords, w = rem % kTileWords;
        const int gr = (row0 + r < H) ? row0 + r : H - 1;
        const int gw = (wbase + w < W32) ? wbase + w : W32 - 1;
        const long fr = (plane && disc_static) ? 0 : f;     // a frame-independent disc is stored once
        const uint32_t v = (plane ? disc : stroke)[(fr * H + gr) * W32 + gw];
        sm[idx] = v;
        any |= v;
    }
    const bool busy = __ballot(any != 0) != 0;
    const int i = chunk * 64 + lane;
    if (i >= nu4) return;
    const int c0 = (16 * i) / 3, ph = i % 3;
    const uint32_t Pbg[6] = {kBg24, kBg24, kBg24, kBg24, kBg24, kBg24};
    const uint4 bgv = pack_rgb16(Pbg, ph);
    uint4* dst = frames + ((long)f * H + row0) * (long)nu4 + i;
    if (!busy) {
        for (int row = row0; row < row1; ++row) {
            aur_store_frame16(dst, bgv);
            dst += nu4;
        }
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t pen = pen_rgb[f] & 0x00ffffffu;
    const int sh = c0 & 31;
    const int ia = (c0 >> 5) - wbase;
    const int ib = (ia + 1 < kTileWords) ? ia + 1 : ia;      // bits of a word past the tile are never needed
    for (int row = row0; row < row1; ++row) {
        const uint32_t* ms = sm + (row - row0) * kTileWords;
        const uint32_t* md = ms + kRowsPerTask * kTileWords;
        const uint64_t sb = (((uint64_t)ms[ib] << 32) | ms[ia]) >> sh;
        const uint64_t db = (((uint64_t)md[ib] << 32) | md[ia]) >> sh;
        uint4 o = bgv;
        if (((sb | db) & 0x3fu) != 0) {
            uint32_t P[6];
#pragma unroll
            for (int c = 0; c < 6; ++c)
                P[c] = ((db >> c) & 1u) ? kDisc24 : (((sb >> c) & 1u) ? pen : kBg24);   // disc is painted last
            o = pack_rgb16(P, ph);
        }
        aur_store_frame16(dst, o);
        dst += nu4;
    }
}

// ---- two-pass plane fill (W % 32 == 0): linear background / disc fill, then patches where the stroke is ----------------
//
// The frames are written at the rate of the linear span fill (raster.hip: one 4 KB-aligned 4 KB piece per workgroup, one
// non-temporal 16-byte store per lane) instead of the 16-row tiles above, whose per-tile mask loads come cold from HBM in
// front of the stores (5.45 TB/s).  What a lane must know in pass 1 is small enough to stay in L2: the centre disc is one
// pixel interval per row (disc_rows_kernel: first / last set bit of the row; a row whose bits are not contiguous is marked
// and its disc pixels go through the patches too), 4 bytes per frame and row, one table for all frames when the disc is
// static (the export default).  Pass 2 walks the stroke plane once, one lane per 32-pixel word, and rewrites the six
// 16-byte columns (32 pixels = 96 bytes exactly) of the few words that have a bit set.
constexpr uint32_t kRowMulti = 0xffffffffu;      // x0 = x1 = 65535: no pixel column of a frame lies inside, pass 1 paints background

__global__ __launch_bounds__(256) void disc_rows_kernel(const uint32_t* __restrict__ disc, int H, int W32,
                                                        uint32_t* __restrict__ rows) {
    // one wave per row
    const int f = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= H) return;
    const uint32_t* src = disc + ((long)f * H + row) * W32;
    int first = 1 << 30, last = -1, count = 0;
    for (int w = lane; w < W32; w += 64) {
        const uint32_t v = src[w];
        if (v) {
            const int lo = 32 * w + __builtin_ctz(v), hi = 32 * w + 31 - __builtin_clz(v);
            first = lo < first ? lo : first;
            last = hi > last ? hi : last;
            count += __builtin_popcount(v);
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int of = __shfl_xor(first, m, 64), ol = __shfl_xor(last, m, 64);
        first = of < first ? of : first;
        last = ol > last ? ol : last;
        count += __shfl_xor(count, m, 64);
    }
    if (lane == 0) {
        uint32_t t = 0u;
        if (count > 0) t = (count == last - first + 1) ? ((uint32_t)first | ((uint32_t)(last + 1) << 16)) : kRowMulti;
        rows[(long)f * H + row] = t;
    }
}

// Pen pixels straight into the frame (aur_raster_closed_path): what pass 2 would patch in from a stroke plane, without the
// plane.  The disc is painted last in the reference, so a run leaves out the row's disc interval (or, in a row whose disc
// is not one interval, every pixel whose disc bit is set).  All primitives paint the same colour: overlaps need no order.
struct PaintSink {
    uint8_t* frame;                  // this frame, uint8 [H][W][3]
    const uint32_t* rows;            // this frame's disc row table
    const uint32_t* disc;            // this frame's disc plane (only read in kRowMulti rows)
    int W, W32;
    uint32_t pen;
    __device__ __forceinline__ void span(long r, int c0, int c1) const {
        if (c0 < 0) c0 = 0;
        if (c1 > W) c1 = W;
        if (c1 <= c0) return;
        const uint32_t t = rows[r];
        const int x0 = (int)(t & 0xffffu), x1 = (int)(t >> 16);
        uint8_t* p = frame + ((long)r * W + c0) * 3;
        for (int c = c0; c < c1; ++c, p += 3) {
            if (t == kRowMulti) {
                if ((disc[r * W32 + (c >> 5)] >> (c & 31)) & 1u) continue;
            } else if (c >= x0 && c < x1) {
                continue;
            }
            p[0] = (uint8_t)pen;
            p[1] = (uint8_t)(pen >> 8);
            p[2] = (uint8_t)(pen >> 16);
        }
    }
};

#ifndef AUR_PAINT_LANES
#define AUR_PAINT_LANES 8          // A/B build knob: lanes that share one stroke primitive
#endif
constexpr int kPaintLanes = AUR_PAINT_LANES;
// Blocks [0, stroke_blocks) of a frame paint the stroke; the blocks behind them paint the disc pixels of the rows pass 1
// could not describe by one interval (kRowMulti), one lane per row: a lane returns at once for every other row -- all
// rows of the discs an export produces.  Stroke pixels leave the disc's pixels out, so the two parts touch disjoint
// pixels and need no order between them.
__global__ __launch_bounds__(256) void paint_stroke_closed_kernel(const double* __restrict__ verts, int nv, int W, int H,
                                                                  int W32, const uint32_t* __restrict__ rows,
                                                                  const uint32_t* __restrict__ disc, int disc_static,
                                                                  const uint32_t* __restrict__ pen_rgb, int stroke_blocks,
                                                                  uint8_t* __restrict__ frames) {
    const int f = blockIdx.y;
    const long fr = disc_static ? 0 : f;
    if ((int)blockIdx.x >= stroke_blocks) {
        const int row = ((int)blockIdx.x - stroke_blocks) * 256 + threadIdx.x;
        if (row >= H || rows[fr * H + row] != kRowMulti) return;
        const uint32_t* src = disc + (fr * H + row) * W32;
        uint8_t* dst = frames + ((long)f * H + row) * (long)W * 3;
        for (int w = 0; w < W32; ++w) {
            uint32_t bits = src[w];
            while (bits) {
                const int x = 32 * w + __builtin_ctz(bits);
                bits &= bits - 1u;
                dst[3 * x] = (uint8_t)kDisc24;
                dst[3 * x + 1] = (uint8_t)(kDisc24 >> 8);
                dst[3 * x + 2] = (uint8_t)(kDisc24 >> 16);
            }
        }
        return;
    }
    const PaintSink sink = {frames + (long)f * H * W * 3, rows + fr * H, disc + fr * H * W32, W, W32,
                            pen_rgb[f] & 0x00ffffffu};
    // kPaintLanes lanes per primitive, each taking every kPaintLanes-th of its pixel rows: the walk over a primitive's rows
    // is a chain of dependent float64 divisions, and one lane per primitive left the kernel latency-bound
    const int t = blockIdx.x * 256 + threadIdx.x;
    stroke_primitive(verts + (long)f * nv * 2, nv, t / kPaintLanes, H, 1.0, 0.0, 0.0, 1, 0, sink, t % kPaintLanes,
                     kPaintLanes);
}

template <bool BASE>
__global__ __launch_bounds__(256) void raster_fill_rows_kernel(const uint32_t* __restrict__ rows, int rows_static, int H,
                                                               int nu4, int upf, float inv_upf, float inv_nu4, int head,
                                                               unsigned total, const uint4* __restrict__ base,
                                                               uint4* __restrict__ frames) {
    const unsigned gi = blockIdx.x * 256u + threadIdx.x;
    if (gi < (unsigned)head) return;
    const unsigned g = gi - (unsigned)head;
    if (g >= total) return;
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
    const int fr = rows_static ? 0 : f;
    const uint32_t t = rows[(long)fr * H + row];        // kRowMulti: this row's disc is painted by the patch pass
    const int x0 = (int)(t & 0xffffu), x1 = (int)(t >> 16);
    const int ph = col % 3, c0 = (16 * col) / 3;
    constexpr uint32_t D0 = kBg24 | (kBg24 << 24), D1 = (kBg24 >> 8) | (kBg24 << 16), D2 = (kBg24 >> 16) | (kBg24 << 8);
    uint4 o;
    if (BASE) {                                          // aur_set_background: the frame starts from the composed image
        o = base[u];
    } else {
        o.x = ph == 0 ? D0 : (ph == 1 ? D1 : D2);
        o.y = ph == 0 ? D1 : (ph == 1 ? D2 : D0);
        o.z = ph == 0 ? D2 : (ph == 1 ? D0 : D1);
        o.w = o.x;
    }
    if (c0 < x1 && c0 + 6 > x0) {
        uint32_t P[6];
        if (BASE) {
            uint32_t M[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                M[c] = (c0 + c >= x0 && c0 + c < x1) ? 0x00ffffffu : 0u;
                P[c] = kDisc24 & M[c];
            }
            const uint4 mk = pack_rgb16(M, ph), pv = pack_rgb16(P, ph);
            o.x = (o.x & ~mk.x) | pv.x;
            o.y = (o.y & ~mk.y) | pv.y;
            o.z = (o.z & ~mk.z) | pv.z;
            o.w = (o.w & ~mk.w) | pv.w;
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) P[c] = (c0 + c >= x0 && c0 + c < x1) ? kDisc24 : kBg24;
            o = pack_rgb16(P, ph);
        }
    }
    aur_store_frame16(frames + g, o);
}

// Rewrites the six 16-byte columns of plane word i of frame f (32 pixels = 96 bytes) when a bit of it must be painted.
// rmw: pass 1 wrote a background image under the word, so the pixels that are not painted keep what the frame holds.
__device__ __forceinline__ void patch_word(uint32_t sb, const uint32_t* __restrict__ disc_frame, bool disc_here, long i,
                                           uint32_t pen, uint4* __restrict__ frame, bool rmw) {
    uint32_t db = 0u;
    if (disc_here || sb != 0u) db = disc_frame[i];                   // the disc is painted last: it covers the stroke
    if (sb == 0u && !(disc_here && db != 0u)) return;
    uint4* dst = frame + i * 6;
    const uint32_t need = disc_here ? (sb | db) : sb;                 // pass 1 painted everything else of the word
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int c0 = (16 * k) / 3;                                  // 0, 5, 10, 16, 21, 26
        if (((need >> c0) & 0x3fu) == 0u) continue;                   // this 16-byte column is already right
        uint32_t P[6];
        if (rmw) {
            uint32_t M[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int b = c0 + c;                                 // <= 31
                M[c] = ((need >> b) & 1u) ? 0x00ffffffu : 0u;
                P[c] = (((db >> b) & 1u) ? kDisc24 : pen) & M[c];
            }
            const uint4 mk = pack_rgb16(M, k % 3), pv = pack_rgb16(P, k % 3);
            uint4 o = dst[k];
            o.x = (o.x & ~mk.x) | pv.x;
            o.y = (o.y & ~mk.y) | pv.y;
            o.z = (o.z & ~mk.z) | pv.z;
            o.w = (o.w & ~mk.w) | pv.w;
            dst[k] = o;
            continue;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int b = c0 + c;                                     // <= 31
            P[c] = ((db >> b) & 1u) ? kDisc24 : (((sb >> b) & 1u) ? pen : kBg24);
        }
        dst[k] = pack_rgb16(P, k % 3);
    }
}

// VEC: one lane reads four stroke words with one 16-byte load (words % 4 == 0, 16-byte aligned plane); almost all of
// them are zero, so the pass is a stream over the stroke plane.
template <bool VEC>
__global__ __launch_bounds__(256) void raster_patch_planes_kernel(const uint32_t* __restrict__ stroke,
                                                                  const uint32_t* __restrict__ disc,
                                                                  const uint32_t* __restrict__ rows,
                                                                  const uint32_t* __restrict__ pen_rgb, int words,
                                                                  int W32, int H, int disc_static, int rmw,
                                                                  uint4* __restrict__ frames) {
    // words = H * W32 per frame; word i of frame f covers the uint4 columns 6 i .. 6 i + 5 of the frame (nu4 = 6 W32)
    const int f = blockIdx.y;
    const int fr = disc_static ? 0 : f;
    const uint32_t* rf = rows + (long)fr * H;                         // kRowMulti: the row's disc is not in pass 1
    const uint32_t* sf = stroke + (long)f * words;
    const uint32_t* df = disc + (long)fr * words;
    uint4* frame = frames + (long)f * words * 6;
    const uint32_t pen = pen_rgb[f] & 0x00ffffffu;
    if (VEC) {
        const int i4 = blockIdx.x * 256 + threadIdx.x;
        if (4 * i4 >= words) return;
        const uint4 s4 = reinterpret_cast<const uint4*>(sf)[i4];
        const bool disc_here = rf[(4 * i4) / W32] == kRowMulti;       // W32 % 4 == 0: the four words share a row
        if (!disc_here && (s4.x | s4.y | s4.z | s4.w) == 0u) return;
        patch_word(s4.x, df, disc_here, 4L * i4, pen, frame, rmw != 0);
        patch_word(s4.y, df, disc_here, 4L * i4 + 1, pen, frame, rmw != 0);
        patch_word(s4.z, df, disc_here, 4L * i4 + 2, pen, frame, rmw != 0);
        patch_word(s4.w, df, disc_here, 4L * i4 + 3, pen, frame, rmw != 0);
    } else {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= words) return;
        patch_word(sf[i], df, rf[i / W32] == kRowMulti, i, pen, frame, rmw != 0);
    }
}

// plane &= ~(excl_a | excl_b), and the whole plane of a frame is cleared when enable[f] == 0.
__global__ __launch_bounds__(256) void plane_exclude_kernel(uint32_t* __restrict__ plane,
                                                            const uint32_t* __restrict__ excl_a,
                                                            const uint32_t* __restrict__ excl_b, int b_static,
                                                            const uint8_t* __restrict__ enable, int words) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    const size_t o = (size_t)f * words + i;
    uint32_t v = plane[o];
    if (enable != nullptr && enable[f] == 0) v = 0;
    if (excl_a != nullptr) v &= ~excl_a[o];
    if (excl_b != nullptr) v &= ~excl_b[b_static ? (size_t)i : o];
    plane[o] = v;
}

// Any width (W not a multiple of 16, or an unaligned batch): one thread per pixel, three byte stores.  Same colours
// and painting order as raster_fill_planes_kernel; only odd export sizes such as 1366x768 come here.
__global__ __launch_bounds__(256) void raster_fill_planes_px_kernel(const uint32_t* __restrict__ stroke,
                                                                    const uint32_t* __restrict__ disc,
                                                                    const uint32_t* __restrict__ pen_rgb, int W, int H,
                                                                    int W32, int disc_static,
                                                                    const uint8_t* __restrict__ base,
                                                                    uint8_t* __restrict__ frames) {
    const int f = blockIdx.z, row = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const uint32_t sb = (stroke[((long)f * H + row) * W32 + (x >> 5)] >> (x & 31)) & 1u;
    const uint32_t db = (disc[((long)(disc_static ? 0 : f) * H + row) * W32 + (x >> 5)] >> (x & 31)) & 1u;
    uint32_t c = kBg24;
    if (base != nullptr) {
        const uint8_t* b = base + ((long)row * W + x) * 3;
        c = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16);
    }
    c = db ? kDisc24 : (sb ? (pen_rgb[f] & 0x00ffffffu) : c);
    uint8_t* px = frames + (((size_t)f * H + row) * W + x) * 3;
    px[0] = (uint8_t)c;
    px[1] = (uint8_t)(c >> 8);
    px[2] = (uint8_t)(c >> 16);
}

}  // namespace

extern "C" int aur_radial_stage1(aur_ctx* ctx, const float* d_logspec, int n_frames, int mirror, int wave_smoothness,
                                 double* d_vals1, int L1, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0, "aur_radial_stage1: n_frames %d < 0", n_frames);
    AUR_REQUIRE(wave_smoothness >= 0 && wave_smoothness <= 100, "aur_radial_stage1: smoothness %d outside [0, 100]",
                wave_smoothness);
    const int L0 = mirror ? 2 * AUR_N_BARS - 2 : AUR_N_BARS;
    const int up = (wave_smoothness > 0) ? 1 + wave_smoothness / 25 : 1;
    AUR_REQUIRE(L1 == L0 * up, "aur_radial_stage1: L1 %d != %d", L1, L0 * up);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_logspec && d_vals1, "aur_radial_stage1: null device pointer");
    const double step = (double)(L0 - 1) / (double)(L0 * up - 1);     // np.linspace(0, L - 1, L * up) step
    hipLaunchKernelGGL(radial_stage1_kernel, dim3(n_frames), dim3(256), 0, (hipStream_t)stream, d_logspec, mirror,
                       wave_smoothness, step, L0, L1, d_vals1);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_radial_ema(aur_ctx* ctx, double* d_vals1, int n_frames, int L1, double alpha, int has_state,
                              double* d_state, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && L1 > 0 && L1 <= kMaxL / 4, "aur_radial_ema: bad shape F=%d L1=%d", n_frames, L1);
    AUR_REQUIRE(alpha > 0.0 && alpha <= 0.95, "aur_radial_ema: alpha %g outside (0, 0.95]", alpha);
    AUR_REQUIRE(d_vals1 && d_state, "aur_radial_ema: null device pointer");
    hipLaunchKernelGGL(radial_ema_kernel, dim3((L1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_vals1, n_frames,
                       L1, alpha, has_state, d_state);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_radial_vertices(aur_ctx* ctx, const double* d_vals1, const double* d_energy, int n_frames, int L1,
                                   int smooth_amount, const double* d_cos_sin, int W, int H, double* d_verts, int L2,
                                   void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && L1 > 0 && L1 <= kMaxL / 4, "aur_radial_vertices: bad shape F=%d L1=%d", n_frames, L1);
    AUR_REQUIRE(smooth_amount >= 0 && smooth_amount <= 100, "aur_radial_vertices: amount %d outside [0, 100]",
                smooth_amount);
    const int up = (L1 > 3 && smooth_amount > 0) ? 1 + smooth_amount / 25 : 1;
    AUR_REQUIRE(L2 == L1 * up && L2 <= kMaxL, "aur_radial_vertices: L2 %d != %d", L2, L1 * up);
    AUR_REQUIRE(W >= 16 && H >= 16, "aur_radial_vertices: frame %dx%d too small", W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_vals1 && d_energy && d_cos_sin && d_verts, "aur_radial_vertices: null device pointer");
    const double step = (double)(L1 - 1) / (double)(L1 * up - 1);
    const int lds = (L1 + 2 * L2) * (int)sizeof(double);
    AUR_CHECK_HIP(hipFuncSetAttribute((const void*)radial_vertices_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      lds));
    hipLaunchKernelGGL(radial_vertices_kernel, dim3(n_frames), dim3(256), lds, (hipStream_t)stream, d_vals1, d_energy,
                       L1, smooth_amount, step, L2, d_cos_sin, W, H, d_verts);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_envelope_vertices(aur_ctx* ctx, const float* d_logspec, const double* d_energy,
                                     const double* d_phase, int n_frames, const aur_envelope_params* p,
                                     const double* d_cos_sin360, int W, int H, double* d_verts, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(p != nullptr, "aur_envelope_vertices: null params");
    AUR_REQUIRE(n_frames >= 0, "aur_envelope_vertices: n_frames %d < 0", n_frames);
    AUR_REQUIRE(W >= 16 && H >= 16, "aur_envelope_vertices: frame %dx%d too small", W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_logspec && d_energy && d_phase && d_cos_sin360 && d_verts, "aur_envelope_vertices: null device pointer");
    EnvArgs a;
    a.mirror = p->radial_mirror ? 1 : 0;
    a.W = W;
    a.H = H;
    a.noise_amt = (double)p->edge_waviness / 100.0;
    a.audio_amt = p->feather_audio_enabled ? (double)p->feather_audio_amount / 100.0 : 0.0;
    a.sens_ratio = p->feather_sensitivity / fmax(1e-3, p->waveform_sensitivity);
    a.rot = p->rotation_rad;
    hipLaunchKernelGGL(envelope_vertices_kernel, dim3(n_frames), dim3(384), 0, (hipStream_t)stream, d_logspec, d_energy,
                       d_phase, a, d_cos_sin360, d_verts);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

// ---- plane blend --------------------------------------------------------------------------------------------------
// Source-over of K constant colours (r, g, b, a), plane k wherever its bit is set, in plane order, into finished RGB24
// frames.  Used for the feather rings: `p.fillPath(feather_path, col)` for k = 1..6 with col = QColor(13, 13, 18,
// max(10, 70 - 10 k)) (:1099-1117), no antialiasing (the hint was restored off at :1096); a = 255 overwrites (a disc
// composited over a frame of another mode).  Arithmetic = Qt's premultiplied ARGB32 source-over,
//   d' = div255(c a) + div255(d (255 - a)),   div255(x) = (x + (x >> 8) + 0x80) >> 8      (parity-unpinned vs Qt).
constexpr int kBlendMaxK = 8;
struct BlendColours {
    uint32_t pm[kBlendMaxK][3];      // premultiplied source div255(c a), per channel (computed on the host)
    uint32_t inv[kBlendMaxK];        // 255 - a
    uint32_t pm_rb[kBlendMaxK];      // pm[k][0] | pm[k][2] << 16
    // Planes from the last opaque one on decide a pixel alone when all of them cover it (whatever lay there is
    // overwritten, then blended over in a fixed sequence): that colour, as the three dwords of eight such pixels.
    int last_opaque;                 // -1: no plane has alpha 255
    uint32_t solid[3];
};
__host__ __device__ __forceinline__ uint32_t div255(uint32_t x) { return (x + (x >> 8) + 0x80u) >> 8; }

// Each lane first reads the K plane words of one 32-pixel word of the frame (64 consecutive words per wave); a wave
// whose 2048 pixels have no bit set (most of a frame) ends there.  Otherwise the words are parked in LDS, the
// quarter words (8 pixels) with any bit set are queued, and the wave works the queue off 64 entries at a time, one
// lane per 8 pixels, so the blend passes run with full lanes.  The blend arithmetic is what bounds the kernel (K x 3 channels per covered pixel), so K is a
// template parameter (KT >= K planes are looped over) and R and B share one register: the two 16-bit products of
// (R | B << 16) * (255 - a) and their div255 never carry into each other.
// VEC: the 24 bytes of the 8 pixels are 8-byte aligned (W a multiple of 8 and an 8-byte aligned batch): three uint2
// loads, bytes blended in registers, three uint2 stores; otherwise byte accesses.
template <int KT>
__device__ __forceinline__ void blend_pixel(const uint32_t (&m)[KT], int b, const BlendColours& col, uint32_t& rb,
                                            uint32_t& g) {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        if ((m[k] >> b) & 1u) {
            // operands are below 2^24 / 2^8: the 24-bit multiply is exact and runs at full rate
            const uint32_t x = __umul24(rb, col.inv[k]);
            rb = col.pm_rb[k] + (((x + ((x >> 8) & 0x00ff00ffu) + 0x00800080u) >> 8) & 0x00ff00ffu);
            g = col.pm[k][1] + div255(__umul24(g, col.inv[k]));
        }
    }
}

// Read-modify-write of the covered pixels (bits of `any`) of one quarter word: op(b, rb, g) updates pixel b held as
// R | B << 16 and G.
template <bool VEC, class Op>
__device__ __forceinline__ void quarter_rmw(uint32_t any, const Op& op, int W, int x0, uint8_t* __restrict__ px) {
    if (VEC && x0 + 8 <= W) {
        uint2* q2 = reinterpret_cast<uint2*>(px);
        uint32_t d[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint2 v = q2[i];
            d[2 * i] = v.x;
            d[2 * i + 1] = v.y;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if ((any >> b) & 1u) {
                // bytes 3b, 3b + 1, 3b + 2 of the 24: v_perm_b32 picks from {S0 (selectors 4-7), S1 (0-3)}, 0x0c = zero
                const int ir = (3 * b) >> 2, orr = (3 * b) & 3;
                const int ig = (3 * b + 1) >> 2, og = (3 * b + 1) & 3;
                const int ib = (3 * b + 2) >> 2, ob = (3 * b + 2) & 3;
                uint32_t rb = __builtin_amdgcn_perm(d[ib], d[ir], (uint32_t)orr | 0x0c000c00u | ((uint32_t)(4 + ob) << 16));
                uint32_t g = (d[ig] >> (8 * og)) & 0xffu;
                op(b, rb, g);
                const uint32_t keep = 0x03020100u;
                d[ir] = __builtin_amdgcn_perm(rb, d[ir], (keep & ~(0xffu << (8 * orr))) | (4u << (8 * orr)));
                d[ig] = __builtin_amdgcn_perm(g, d[ig], (keep & ~(0xffu << (8 * og))) | (4u << (8 * og)));
                d[ib] = __builtin_amdgcn_perm(rb, d[ib], (keep & ~(0xffu << (8 * ob))) | (6u << (8 * ob)));
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) q2[i] = uint2{d[2 * i], d[2 * i + 1]};
        return;
    }
    while (any) {
        const int b = __builtin_ctz(any);
        any &= any - 1;
        if (x0 + b >= W) break;
        uint32_t rb = (uint32_t)px[3 * b] | ((uint32_t)px[3 * b + 2] << 16), g = px[3 * b + 1];
        op(b, rb, g);
        px[3 * b] = (uint8_t)rb;
        px[3 * b + 1] = (uint8_t)g;
        px[3 * b + 2] = (uint8_t)(rb >> 16);
    }
}

template <bool VEC, int KT>
// Hp, W32p, xw0, y0: the planes may cover only a box of the frame (aur_blend_planes_box): Hp rows of W32p words whose first
// word is word xw0 of frame row y0.
__global__ __launch_bounds__(256) void blend_planes_kernel(const uint32_t* __restrict__ planes, int K, BlendColours col,
                                                           int W, int H, int Hp, int W32p, int xw0, int y0,
                                                           uint8_t* __restrict__ frames) {
    __shared__ uint32_t s_w[4][KT + 1][64];
    __shared__ uint8_t s_q[4][256];
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int words = Hp * W32p;
    const int word0 = blockIdx.x * 256 + wave * 64;         // first word of this wave
    if (word0 >= words) return;                             // wave-uniform
    const int mine = (word0 + lane < words) ? word0 + lane : words - 1;
    uint32_t any = 0, raw[KT];
    // all loads first, unconditional (planes beyond K re-read plane K - 1): behind a branch, or with the LDS store of
    // one value ahead of the next load, each load waits for the one before it
#pragma unroll
    for (int k = 0; k < KT; ++k) raw[k] = planes[((size_t)f * K + min(k, K - 1)) * words + mine];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const uint32_t v = (k >= K || word0 + lane >= words) ? 0u : raw[k];
        s_w[wave][k][lane] = v;
        any |= v;
    }
    if (__ballot(any != 0) == 0) return;                    // wave-uniform
    s_w[wave][KT][lane] = any;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the non-empty quarter words of the wave's 256, compacted so that the blend passes run with full lanes
    int n = 0;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int id = pass * 64 + lane;                    // word id >> 2 of the wave, quarter id & 3
        const bool on = ((s_w[wave][KT][id >> 2] >> (8 * (id & 3))) & 0xffu) != 0;
        const uint64_t bal = __ballot(on);
        const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (on) s_q[wave][pos] = (uint8_t)id;
        n += __popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
    for (int base = 0; base < n; base += 64) {
        if (base + lane >= n) continue;
        const int id = s_q[wave][base + lane];
        const int src = id >> 2, q = id & 3;
        const uint32_t a8 = (s_w[wave][KT][src] >> (8 * q)) & 0xffu;
        const int word = word0 + src;
        const int row = word / W32p, w = word - row * W32p;
        uint32_t m[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) m[k] = (s_w[wave][k][src] >> (8 * q)) & 0xffu;
        const int x0 = (w + xw0) * 32 + q * 8;
        if (VEC && col.last_opaque >= 0 && x0 + 8 <= W) {
            // the disc's interior: every plane from the opaque one on covers all eight pixels -- stores only
            uint32_t all = 0xffu;
#pragma unroll
            for (int k = 0; k < KT; ++k)
                if (k >= col.last_opaque && k < K) all &= m[k];
            if (all == 0xffu) {
                uint2* q2 = reinterpret_cast<uint2*>(frames + (((size_t)f * H + (row + y0)) * W + x0) * 3);
                q2[0] = uint2{col.solid[0], col.solid[1]};
                q2[1] = uint2{col.solid[2], col.solid[0]};
                q2[2] = uint2{col.solid[1], col.solid[2]};
                continue;
            }
        }
        quarter_rmw<VEC>(a8, [&](int b, uint32_t& rb, uint32_t& g) { blend_pixel<KT>(m, b, col, rb, g); }, W, x0,
                         frames + (((size_t)f * H + (row + y0)) * W + x0) * 3);
    }
}

template <int KT>
static void launch_blend(bool vec, dim3 grid, hipStream_t st, const uint32_t* planes, int K, const BlendColours& col,
                         int W, int H, int Hp, int W32p, int xw0, int y0, uint8_t* frames) {
    if (vec)
        hipLaunchKernelGGL((blend_planes_kernel<true, KT>), grid, dim3(256), 0, st, planes, K, col, W, H, Hp, W32p, xw0, y0,
                           frames);
    else
        hipLaunchKernelGGL((blend_planes_kernel<false, KT>), grid, dim3(256), 0, st, planes, K, col, W, H, Hp, W32p, xw0,
                           y0, frames);
}

// ---- additive planes (glow) ----------------------------------------------------------------------------------------
// `_draw_glow_offscreen` (rt_widget.py:634-653): up to kAddMaxK strokes of the same path with pens of growing width and
// falling alpha, CompositionMode_Plus.  Layer k of frame f adds its premultiplied colour pm[f][k] (r | g << 8 | b << 16,
// div255(c a) computed on the host from the frame's energy) to every pixel of its plane, each channel saturating at 255
// (Qt's comp_func_Plus on premultiplied ARGB32 with an opaque destination; saturating after every layer equals saturating
// the sum).  Planes are layer-major, uint32 [K][F][H][W32], so that a layer is one aur_mask_stroke_path call.
// One lane per 32-pixel word; lanes whose word has no bit in any layer (most of a frame) stop after the K loads.
constexpr int kAddMaxK = 16;
// One lane per quarter word (8 pixels): the four lanes of a word read the same plane words (one cache line, no extra HBM
// traffic) and a lane whose 8 pixels have no bit in any layer (most of a frame) stops after the K loads.  The stroke and
// disc planes that the reference paints afterwards are masked out here (excl_a per frame, excl_b per frame or static).
template <bool VEC, int KT>
__global__ __launch_bounds__(256) void add_planes_kernel(const uint32_t* __restrict__ planes, int K, size_t layer_stride,
                                                         const uint32_t* __restrict__ pm,
                                                         const uint32_t* __restrict__ excl_a,
                                                         const uint32_t* __restrict__ excl_b, int excl_b_static, int W,
                                                         int H, int W32, uint8_t* __restrict__ frames) {
    const int f = blockIdx.y;
    const int words = H * W32;
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int word = id >> 2, q = id & 3;
    if (word >= words) return;
    const size_t at = (size_t)f * words + word;
    // every load is unconditional (layers beyond K re-read layer K - 1, a missing exclusion plane reads layer 0) so that
    // all of them are in flight at once: behind `k < K` branches each load waited for the one before it
    const uint32_t* pa = excl_a ? excl_a + at : planes + at;
    const uint32_t* pb = excl_b ? excl_b + (excl_b_static ? (size_t)word : at) : planes + at;
    uint32_t va = *pa, vb = *pb, raw[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) raw[k] = planes[(size_t)min(k, K - 1) * layer_stride + at];
    if (!excl_a) va = 0;
    if (!excl_b) vb = 0;
    const uint32_t keep = (~(va | vb) >> (8 * q)) & 0xffu;
    uint32_t m[KT], any = 0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        m[k] = k < K ? (raw[k] >> (8 * q)) & keep : 0u;
        any |= m[k];
    }
    if (any == 0) return;
    uint32_t crb[KT], cg[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const uint32_t c = k < K ? pm[(size_t)f * K + k] : 0u;      // block-uniform
        crb[k] = c & 0x00ff00ffu;
        cg[k] = (c >> 8) & 0xffu;
    }
    const int row = word / W32, w = word - row * W32;
    const int x0 = w * 32 + q * 8;
    quarter_rmw<VEC>(any, [&](int b, uint32_t& rb, uint32_t& g) {
        uint32_t srb = rb, sg = g;                                   // at most 255 + 16 * 255 per 16-bit half
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const uint32_t on = 0u - ((m[k] >> b) & 1u);
            srb += crb[k] & on;
            sg += cg[k] & on;
        }
        const uint32_t r = min(srb & 0xffffu, 255u), bl = min(srb >> 16, 255u);
        rb = r | (bl << 16);
        g = min(sg, 255u);
    }, W, x0, frames + (((size_t)f * H + row) * W + x0) * 3);
}

// CompositionMode_Multiply of a constant colour on the pixels of one plane (`_apply_fill` with radial_fill_blend =
// "multiply" over a background image, rt_widget.py:859-862): Qt's multiply_op on premultiplied ARGB32 with an opaque
// destination is d' = div255(s d + d (255 - a)) = div255(d (s + 255 - a)) per channel, s = div255(c a) <= a, so the
// per-channel factor s + 255 - a (at most 255) comes from the host.  One lane per quarter word.
template <bool VEC>
__global__ __launch_bounds__(256) void multiply_plane_kernel(const uint32_t* __restrict__ plane, uint32_t fr, uint32_t fg,
                                                             uint32_t fb, int W, int H, int W32,
                                                             uint8_t* __restrict__ frames) {
    const int f = blockIdx.y;
    const int words = H * W32;
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int word = id >> 2, q = id & 3;
    if (word >= words) return;
    const uint32_t m = (plane[(size_t)f * words + word] >> (8 * q)) & 0xffu;
    if (m == 0u) return;
    const int row = word / W32, w = word - row * W32;
    const int x0 = w * 32 + q * 8;
    quarter_rmw<VEC>(m, [&](int b, uint32_t& rb, uint32_t& g) {
        const uint32_t r = div255(__umul24(rb & 0xffffu, fr)), bl = div255(__umul24(rb >> 16, fb));
        rb = r | (bl << 16);
        g = div255(__umul24(g, fg));
    }, W, x0, frames + (((size_t)f * H + row) * W + x0) * 3);
}

template <int KT>
static void launch_add(bool vec, dim3 grid, hipStream_t st, const uint32_t* planes, int K, size_t layer_stride,
                       const uint32_t* pm, const uint32_t* ea, const uint32_t* eb, int eb_static, int W, int H, int W32,
                       uint8_t* frames) {
    if (vec)
        hipLaunchKernelGGL((add_planes_kernel<true, KT>), grid, dim3(256), 0, st, planes, K, layer_stride, pm, ea, eb,
                           eb_static, W, H, W32, frames);
    else
        hipLaunchKernelGGL((add_planes_kernel<false, KT>), grid, dim3(256), 0, st, planes, K, layer_stride, pm, ea, eb,
                           eb_static, W, H, W32, frames);
}

extern "C" int aur_wave_linear_vertices(aur_ctx* ctx, const float* d_samples, const int32_t* d_col2samp,
                                        const double* d_energy, int n_frames, int W, int H, double* d_verts,
                                        void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_wave_linear_vertices: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= kMaxL + 1 && H >= 16, "aur_wave_linear_vertices: frame %dx%d unsupported (W <= %d)", W, H,
                kMaxL + 1);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_samples && d_col2samp && d_energy && d_verts, "aur_wave_linear_vertices: null device pointer");
    hipLaunchKernelGGL(wave_linear_vertices_kernel, dim3((W + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)stream,
                       d_samples, d_col2samp, d_energy, W, H, d_verts);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_wave_circular_vertices(aur_ctx* ctx, const float* d_samples, const double* d_energy, int n_frames,
                                          const double* d_cos_sin, int W, int H, double* d_verts, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_wave_circular_vertices: n_frames %d outside [0, 65535]",
                n_frames);
    AUR_REQUIRE(W >= 16 && H >= 16, "aur_wave_circular_vertices: frame %dx%d too small", W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_samples && d_energy && d_cos_sin && d_verts, "aur_wave_circular_vertices: null device pointer");
    static_assert(AUR_BLOCK % 256 == 0, "one thread per sample");
    hipLaunchKernelGGL(wave_circular_vertices_kernel, dim3(AUR_BLOCK / 256, n_frames), dim3(256), 0,
                       (hipStream_t)stream, d_samples, d_energy, d_cos_sin, W, H, d_verts);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_feather_ring_vertices(aur_ctx* ctx, const float* d_logspec, const double* d_energy,
                                        const double* d_phase, int n_frames, const aur_envelope_params* p,
                                        const double* d_cos_sin360, int W, int H, double* d_rings, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(p != nullptr, "aur_feather_ring_vertices: null params");
    AUR_REQUIRE(n_frames >= 0, "aur_feather_ring_vertices: n_frames %d < 0", n_frames);
    AUR_REQUIRE(W >= 16 && H >= 16, "aur_feather_ring_vertices: frame %dx%d too small", W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_logspec && d_energy && d_phase && d_cos_sin360 && d_rings,
                "aur_feather_ring_vertices: null device pointer");
    EnvArgs a;
    a.mirror = p->radial_mirror ? 1 : 0;
    a.W = W;
    a.H = H;
    a.noise_amt = (double)p->edge_waviness / 100.0;
    a.audio_amt = p->feather_audio_enabled ? (double)p->feather_audio_amount / 100.0 : 0.0;
    a.sens_ratio = p->feather_sensitivity / fmax(1e-3, p->waveform_sensitivity);
    a.rot = p->rotation_rad;
    hipLaunchKernelGGL(feather_ring_vertices_kernel, dim3(n_frames), dim3(384), 0, (hipStream_t)stream, d_logspec,
                       d_energy, d_phase, a, d_cos_sin360, d_rings);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_blend_planes(aur_ctx* ctx, const uint32_t* d_planes, int K, const uint8_t* h_rgba, int n_frames,
                               int W, int H, uint8_t* d_frames, void* stream) {
    return aur_blend_planes_box(ctx, d_planes, K, h_rgba, n_frames, W, H, 0, 0, W, H, d_frames, stream);
}

extern "C" int aur_blend_planes_box(aur_ctx* ctx, const uint32_t* d_planes, int K, const uint8_t* h_rgba, int n_frames,
                                    int W, int H, int box_x, int box_y, int box_w, int box_h, uint8_t* d_frames,
                                    void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(K >= 1 && K <= kBlendMaxK, "aur_blend_planes: K %d outside [1, %d]", K, kBlendMaxK);
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_blend_planes: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_blend_planes: frame size %dx%d unsupported", W, H);
    AUR_REQUIRE(h_rgba != nullptr, "aur_blend_planes: null colour table");
    AUR_REQUIRE(box_x >= 0 && box_y >= 0 && box_w >= 1 && box_h >= 1 && box_x + box_w <= ((W + 31) & ~31) &&
                    box_y + box_h <= H && box_x % 32 == 0,
                "aur_blend_planes_box: box %d,%d %dx%d outside the %dx%d frame or not on a 32-pixel column", box_x, box_y,
                box_w, box_h, W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_planes && d_frames, "aur_blend_planes: null device pointer");
    BlendColours col = {};
    for (int k = 0; k < K; ++k) {
        const uint32_t a = h_rgba[4 * k + 3];
        for (int ch = 0; ch < 3; ++ch) col.pm[k][ch] = div255((uint32_t)h_rgba[4 * k + ch] * a);
        col.inv[k] = 255u - a;
        col.pm_rb[k] = col.pm[k][0] | (col.pm[k][2] << 16);
    }
    col.last_opaque = -1;
    for (int k = 0; k < K; ++k)
        if (col.inv[k] == 0u) col.last_opaque = k;
    if (col.last_opaque >= 0) {
        uint32_t c[3];
        for (int ch = 0; ch < 3; ++ch) {
            c[ch] = col.pm[col.last_opaque][ch];                     // alpha 255: d' = c
            for (int k = col.last_opaque + 1; k < K; ++k) c[ch] = col.pm[k][ch] + div255(c[ch] * col.inv[k]);
        }
        const uint32_t r = c[0], g = c[1], b = c[2];
        col.solid[0] = r | (g << 8) | (b << 16) | (r << 24);
        col.solid[1] = g | (b << 8) | (r << 16) | (g << 24);
        col.solid[2] = b | (r << 8) | (g << 16) | (b << 24);
    }
    const int W32p = (box_w + 31) / 32, xw0 = box_x / 32;
    const dim3 grid((box_h * W32p + 255) / 256, n_frames);  // one lane per 32-pixel word of the planes
    const bool vec = W % 8 == 0 && ((uintptr_t)d_frames & 7u) == 0;
    hipStream_t st = (hipStream_t)stream;
#define AUR_BLEND(KT) launch_blend<KT>(vec, grid, st, d_planes, K, col, W, H, box_h, W32p, xw0, box_y, d_frames)
    if (K == 1) AUR_BLEND(1);                               // fill, shadow
    else if (K <= 6) AUR_BLEND(6);                          // feather rings
    else if (K == 7) AUR_BLEND(7);                          // disc + rings
    else AUR_BLEND(kBlendMaxK);
#undef AUR_BLEND
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_add_planes(aur_ctx* ctx, const uint32_t* d_planes, int K, const uint8_t* d_pm_rgb,
                             const uint32_t* d_excl_a, const uint32_t* d_excl_b, int excl_b_static, int n_frames, int W,
                             int H, uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(K >= 1 && K <= kAddMaxK, "aur_add_planes: K %d outside [1, %d]", K, kAddMaxK);
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_add_planes: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_add_planes: frame size %dx%d unsupported", W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_planes && d_pm_rgb && d_frames, "aur_add_planes: null device pointer");
    AUR_REQUIRE(((uintptr_t)d_pm_rgb & 3u) == 0, "aur_add_planes: colour table not 4-byte aligned");
    const int W32 = (W + 31) / 32;
    const size_t layer_stride = (size_t)n_frames * H * W32;
    const dim3 grid((4 * H * W32 + 255) / 256, n_frames);   // one lane per quarter word
    const bool vec = W % 8 == 0 && ((uintptr_t)d_frames & 7u) == 0;
    const uint32_t* pm = reinterpret_cast<const uint32_t*>(d_pm_rgb);
    hipStream_t st = (hipStream_t)stream;
    const int ebs = excl_b_static ? 1 : 0;
    if (K <= 8) launch_add<8>(vec, grid, st, d_planes, K, layer_stride, pm, d_excl_a, d_excl_b, ebs, W, H, W32, d_frames);
    else launch_add<kAddMaxK>(vec, grid, st, d_planes, K, layer_stride, pm, d_excl_a, d_excl_b, ebs, W, H, W32, d_frames);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_multiply_plane(aur_ctx* ctx, const uint32_t* d_plane, const uint8_t* h_rgba, int n_frames, int W, int H,
                                  uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_multiply_plane: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_multiply_plane: frame size %dx%d unsupported", W, H);
    AUR_REQUIRE(h_rgba != nullptr, "aur_multiply_plane: null colour");
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_plane && d_frames, "aur_multiply_plane: null device pointer");
    const uint32_t a = h_rgba[3];
    uint32_t fac[3];
    for (int ch = 0; ch < 3; ++ch) fac[ch] = div255((uint32_t)h_rgba[ch] * a) + 255u - a;
    const int W32 = (W + 31) / 32;
    const dim3 grid((4 * H * W32 + 255) / 256, n_frames);
    const bool vec = W % 8 == 0 && ((uintptr_t)d_frames & 7u) == 0;
    hipStream_t st = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(multiply_plane_kernel<true>, grid, dim3(256), 0, st, d_plane, fac[0], fac[1], fac[2], W, H, W32,
                           d_frames);
    else
        hipLaunchKernelGGL(multiply_plane_kernel<false>, grid, dim3(256), 0, st, d_plane, fac[0], fac[1], fac[2], W, H, W32,
                           d_frames);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

static int check_plane_args(const char* who, int n_frames, int nv, int W, int H) {
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "%s: n_frames %d outside [0, 65535]", who, n_frames);
    AUR_REQUIRE(nv >= 3 && nv <= kMaxL + 1, "%s: %d vertices outside [3, %d]", who, nv, kMaxL + 1);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "%s: frame size %dx%d outside [16..16384]^2", who, W, H);
    return AUR_OK;
}

extern "C" int aur_mask_stroke_closed(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H,
                                      uint32_t* d_plane, void* stream) {
    return aur_mask_stroke_path(ctx, d_verts, n_vertices, n_frames, W, H, 1.0, 0.0, 0.0, 1, 0, d_plane, stream);
}

extern "C" int aur_mask_stroke_path(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H,
                                    double half_width, double dx, double dy, int closed, int round_join,
                                    uint32_t* d_plane, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_plane_args("aur_mask_stroke_closed", n_frames, n_vertices, W, H))) return rc;
    AUR_REQUIRE(half_width > 0.0 && half_width <= 256.0, "aur_mask_stroke_closed: half width %g outside (0, 256]",
                half_width);
    AUR_REQUIRE(dx == dx && dy == dy && fabs(dx) <= 65536.0 && fabs(dy) <= 65536.0,
                "aur_mask_stroke_closed: translation (%g, %g) out of range", dx, dy);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_verts && d_plane, "aur_mask_stroke_closed: null device pointer");
    const int W32 = (W + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    AUR_CHECK_HIP(hipMemsetAsync(d_plane, 0, (size_t)n_frames * H * W32 * 4, st));
    hipLaunchKernelGGL(mask_stroke_closed_kernel, dim3((2 * n_vertices + 255) / 256, n_frames), dim3(256), 0, st,
                       d_verts, n_vertices, W, H, W32, half_width, dx, dy, closed ? 1 : 0, round_join ? 1 : 0, d_plane);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_mask_polygon(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H,
                                uint32_t* d_plane, void* stream) {
    return aur_mask_polygon_box(ctx, d_verts, n_vertices, n_frames, W, H, 0, 0, W, H, d_plane, stream);
}

extern "C" int aur_mask_polygon_box(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int frame_w,
                                    int frame_h, int box_x, int box_y, int box_w, int box_h, uint32_t* d_plane,
                                    void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_plane_args("aur_mask_polygon", n_frames, n_vertices, frame_w, frame_h))) return rc;
    AUR_REQUIRE(box_x >= 0 && box_y >= 0 && box_w >= 1 && box_h >= 1 && box_x + box_w <= ((frame_w + 31) & ~31) &&
                    box_y + box_h <= frame_h && box_x % 32 == 0,
                "aur_mask_polygon_box: box %d,%d %dx%d outside the %dx%d frame or not on a 32-pixel column", box_x, box_y,
                box_w, box_h, frame_w, frame_h);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_verts && d_plane, "aur_mask_polygon: null device pointer");
    // the box's right edge never lies beyond the frame's: pixels past the last column stay 0
    const int W = (box_x + box_w > frame_w) ? frame_w - box_x : box_w, H = box_h;
    AUR_REQUIRE(W >= 1, "aur_mask_polygon_box: box starts past the last pixel column");
    const int W32 = (box_w + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    // rows per block: as many as fit the block's LDS bitmap, at most one per thread; a multiple of 4 keeps the uint4
    // zero fill of empty blocks aligned
    int rpb = kPolyWords / W32;
    rpb = rpb > 128 ? 128 : rpb;
    rpb &= ~3;
    const int vec_ok = (W32 % 4 == 0 && ((uintptr_t)d_plane & 15u) == 0) ? 1 : 0;   // rows of empty blocks as uint4
    rpb = rpb < 4 ? 4 : rpb;
    hipLaunchKernelGGL(mask_polygon_kernel, dim3((H + rpb - 1) / rpb, n_frames), dim3(128), 0, st, d_verts, n_vertices, W,
                       H, W32, box_x, box_y, rpb, vec_ok, d_plane);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_plane_exclude(aur_ctx* ctx, uint32_t* d_plane, const uint32_t* d_excl_a, const uint32_t* d_excl_b,
                                int excl_b_static, const uint8_t* d_enable, int n_frames, int W, int H, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_plane_exclude: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_plane_exclude: frame size %dx%d unsupported", W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_plane != nullptr, "aur_plane_exclude: null plane");
    const int words = H * ((W + 31) / 32);
    hipLaunchKernelGGL(plane_exclude_kernel, dim3((words + 255) / 256, n_frames), dim3(256), 0, (hipStream_t)stream,
                       d_plane, d_excl_a, d_excl_b, excl_b_static ? 1 : 0, d_enable, words);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_raster_planes(aur_ctx* ctx, const uint32_t* d_stroke_plane, const uint32_t* d_disc_plane,
                                 int disc_static, const uint8_t* d_pen_rgb, int n_frames, int W, int H,
                                 uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0 && n_frames <= 65535, "aur_raster_planes: n_frames %d outside [0, 65535]", n_frames);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_raster_planes: frame size %dx%d unsupported", W, H);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_stroke_plane && d_disc_plane && d_pen_rgb && d_frames, "aur_raster_planes: null device pointer");
    AUR_REQUIRE(((uintptr_t)d_pen_rgb & 3u) == 0, "aur_raster_planes: misaligned d_pen_rgb");
    const int W32 = (W + 31) / 32;
    if ((rc = aur_check_background(ctx, "aur_raster_planes", W, H))) return rc;
    const long upf_chk = (long)(W * 3 / 16) * H;
    const bool two_pass = W % 32 == 0 && ((uintptr_t)d_frames & 15u) == 0 && ctx->fill_mode != 1 && ctx->d_rows != nullptr &&
                          upf_chk >= 256 && upf_chk < (1L << 26) && (long)n_frames * upf_chk + 256 < (1L << 31) &&
                          (size_t)(disc_static ? 1 : n_frames) * H * 4 <= (size_t)AUR_ROWS_BYTES;
    // odd widths and unaligned batches -- and a background image outside the two-pass fill's shapes -- go pixel by pixel
    if (W % 16 != 0 || ((uintptr_t)d_frames & 15u) != 0 || (ctx->d_base != nullptr && !two_pass)) {
        hipLaunchKernelGGL(raster_fill_planes_px_kernel, dim3((W + 255) / 256, H, n_frames), dim3(256), 0,
                           (hipStream_t)stream, d_stroke_plane, d_disc_plane,
                           reinterpret_cast<const uint32_t*>(d_pen_rgb), W, H, W32, disc_static ? 1 : 0, ctx->d_base,
                           d_frames);
        AUR_CHECK_HIP(hipGetLastError());
        return AUR_OK;
    }
    const int nu4 = W * 3 / 16, nchunks = (nu4 + 63) / 64;
    const long upf_l = (long)nu4 * H;
    const int table_frames = disc_static ? 1 : n_frames;
    if (two_pass) {
        hipStream_t st = (hipStream_t)stream;
        uint32_t* rows = reinterpret_cast<uint32_t*>(ctx->d_rows);
        hipLaunchKernelGGL(disc_rows_kernel, dim3((H + 3) / 4, table_frames), dim3(256), 0, st, d_disc_plane, H, W32, rows);
        const int upf = (int)upf_l;
        const int head = (int)(((uintptr_t)d_frames & 4095u) / 16u);
        const unsigned total = (unsigned)n_frames * (unsigned)upf;
        const unsigned gfill = (unsigned)(((unsigned long)total + (unsigned)head + 255u) / 256u);
        const uint4* base = reinterpret_cast<const uint4*>(ctx->d_base);
        const int rmw = base != nullptr ? 1 : 0;
        if (base != nullptr)
            hipLaunchKernelGGL(raster_fill_rows_kernel<true>, dim3(gfill), dim3(256), 0, st, rows, disc_static ? 1 : 0, H,
                               nu4, upf, 1.0f / (float)upf, 1.0f / (float)nu4, head, total, base,
                               reinterpret_cast<uint4*>(d_frames));
        else
            hipLaunchKernelGGL(raster_fill_rows_kernel<false>, dim3(gfill), dim3(256), 0, st, rows, disc_static ? 1 : 0, H,
                               nu4, upf, 1.0f / (float)upf, 1.0f / (float)nu4, head, total, base,
                               reinterpret_cast<uint4*>(d_frames));
        const int words = H * W32;
        if (W32 % 4 == 0 && ((uintptr_t)d_stroke_plane & 15u) == 0)
            hipLaunchKernelGGL(raster_patch_planes_kernel<true>, dim3((words / 4 + 255) / 256, n_frames), dim3(256), 0, st,
                               d_stroke_plane, d_disc_plane, rows, reinterpret_cast<const uint32_t*>(d_pen_rgb), words,
                               W32, H, disc_static ? 1 : 0, rmw, reinterpret_cast<uint4*>(d_frames));
        else
            hipLaunchKernelGGL(raster_patch_planes_kernel<false>, dim3((words + 255) / 256, n_frames), dim3(256), 0, st,
                               d_stroke_plane, d_disc_plane, rows, reinterpret_cast<const uint32_t*>(d_pen_rgb), words,
                               W32, H, disc_static ? 1 : 0, rmw, reinterpret_cast<uint4*>(d_frames));
        AUR_CHECK_HIP(hipGetLastError());
        return AUR_OK;
    }
    const int ntiles = (H + kRowsPerTask - 1) / kRowsPerTask;
    hipLaunchKernelGGL(raster_fill_planes_kernel, dim3((nchunks * ntiles + 3) / 4, n_frames), dim3(256), 0,
                       (hipStream_t)stream, d_stroke_plane, d_disc_plane, reinterpret_cast<const uint32_t*>(d_pen_rgb), W,
                       H, W32, nu4, nchunks, disc_static ? 1 : 0, reinterpret_cast<uint4*>(d_frames));
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

// A batch of closed-path frames without a stroke plane: pass 1 of aur_raster_planes, then the stroke painted straight into
// the frames.  For renders that need no stroke plane afterwards (no fill / shadow / glow decoration).  Same frames as
// aur_mask_stroke_closed + aur_raster_planes, bit for bit.
extern "C" int aur_disc_rows(aur_ctx* ctx, const uint32_t* d_disc_plane, int n_planes, int W, int H, uint32_t* d_rows,
                             void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_planes >= 0 && n_planes <= 65535, "aur_disc_rows: n_planes %d outside [0, 65535]", n_planes);
    AUR_REQUIRE(W >= 16 && W <= 16384 && H >= 16 && H <= 16384, "aur_disc_rows: frame size %dx%d unsupported", W, H);
    if (n_planes == 0) return AUR_OK;
    AUR_REQUIRE(d_disc_plane && d_rows, "aur_disc_rows: null device pointer");
    hipLaunchKernelGGL(disc_rows_kernel, dim3((H + 3) / 4, n_planes), dim3(256), 0, (hipStream_t)stream, d_disc_plane, H,
                       (W + 31) / 32, d_rows);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_raster_closed_path(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H,
                                      const uint32_t* d_disc_plane, int disc_static, const uint32_t* d_disc_rows,
                                      const uint8_t* d_pen_rgb, uint8_t* d_frames, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    if ((rc = check_plane_args("aur_raster_closed_path", n_frames, n_vertices, W, H))) return rc;
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_verts && d_disc_plane && d_pen_rgb && d_frames, "aur_raster_closed_path: null device pointer");
    AUR_REQUIRE(((uintptr_t)d_pen_rgb & 3u) == 0, "aur_raster_closed_path: misaligned d_pen_rgb");
    if ((rc = aur_check_background(ctx, "aur_raster_closed_path", W, H))) return rc;
    const int W32 = (W + 31) / 32, nu4 = W * 3 / 16;
    const long upf_l = (long)nu4 * H;
    const int table_frames = disc_static ? 1 : n_frames;
    const bool ok = W % 32 == 0 && ((uintptr_t)d_frames & 15u) == 0 && ctx->fill_mode != 1 &&
                    upf_l >= 256 && upf_l < (1L << 26) && (long)n_frames * upf_l + 256 < (1L << 31) &&
                    (d_disc_rows != nullptr ||
                     (ctx->d_rows != nullptr && (size_t)table_frames * H * 4 <= (size_t)AUR_ROWS_BYTES));
    if (!ok) {
        aur_set_error("aur_raster_closed_path: %dx%d frames (or this batch) are outside the linear fill's shapes; use "
                      "aur_mask_stroke_closed + aur_raster_planes", W, H);
        return AUR_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const uint32_t* rows = d_disc_rows;
    if (rows == nullptr) {                       // no table from the caller: built here, in the context's own table
        uint32_t* mine = reinterpret_cast<uint32_t*>(ctx->d_rows);
        hipLaunchKernelGGL(disc_rows_kernel, dim3((H + 3) / 4, table_frames), dim3(256), 0, st, d_disc_plane, H, W32, mine);
        rows = mine;
    }
    const int upf = (int)upf_l;
    const int head = (int)(((uintptr_t)d_frames & 4095u) / 16u);
    const unsigned total = (unsigned)n_frames * (unsigned)upf;
    const unsigned gfill = (unsigned)(((unsigned long)total + (unsigned)head + 255u) / 256u);
    const uint4* base = reinterpret_cast<const uint4*>(ctx->d_base);
    if (base != nullptr)
        hipLaunchKernelGGL(raster_fill_rows_kernel<true>, dim3(gfill), dim3(256), 0, st, rows, disc_static ? 1 : 0, H, nu4,
                           upf, 1.0f / (float)upf, 1.0f / (float)nu4, head, total, base, reinterpret_cast<uint4*>(d_frames));
    else
        hipLaunchKernelGGL(raster_fill_rows_kernel<false>, dim3(gfill), dim3(256), 0, st, rows, disc_static ? 1 : 0, H, nu4,
                           upf, 1.0f / (float)upf, 1.0f / (float)nu4, head, total, base, reinterpret_cast<uint4*>(d_frames));
    const int stroke_blocks = (2 * n_vertices * kPaintLanes + 255) / 256, multi_blocks = (H + 255) / 256;
    hipLaunchKernelGGL(paint_stroke_closed_kernel, dim3(stroke_blocks + multi_blocks, n_frames), dim3(256), 0, st, d_verts,
                       n_vertices, W, H, W32, rows, d_disc_plane, disc_static ? 1 : 0,
                       reinterpret_cast<const uint32_t*>(d_pen_rgb), stroke_blocks, d_frames);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}
