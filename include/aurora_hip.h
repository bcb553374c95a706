/*
 * aurora_hip.h -- C ABI of libaurora_hip.so: the MI355X (gfx950) implementation of Aurora's
 * offline-export hot path (per-frame STFT analysis + per-frame RGB24 rasterisation).
 *
 * The reference (Strepto-Code/Aurora) is pure Python and has no FFI of its own; the boundary it
 * exposes is its Python API (audio/analysis.py, export/exporter.py, export/worker.py).  The host
 * side in aurora_amd/ keeps those signatures and reaches the kernels through the entry points
 * below via ctypes.  Each entry point names the reference code it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *  - every function returns 0 on success, a negative AUR_E* code on failure; aur_last_error()
 *    returns a thread-local message for the last failure on the calling thread;
 *  - nothing throws across the ABI; no torch types; all pointers named d_* are DEVICE pointers
 *    (e.g. torch.Tensor.data_ptr()), h_* are HOST pointers; the caller owns every buffer;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); all launches are
 *    asynchronous on that stream; no entry point except aur_create/aur_destroy/aur_host_* allocates
 *    or synchronises;
 *  - a context (aur_ctx) belongs to one device; use one context per device per thread.
 */
#ifndef AURORA_HIP_H
#define AURORA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUR_OK            0
#define AUR_EINVAL       -1   /* bad argument (shape, size, null pointer, unsupported value) */
#define AUR_EHIP         -2   /* a HIP runtime call failed (message holds hipGetErrorString) */
#define AUR_ENOMEM       -3
#define AUR_EUNSUPPORTED -4

#define AUR_FFT_SIZE   2048   /* audio/analysis.py:6, export/exporter.py:114 (always 2048 in export) */
#define AUR_BLOCK      1024   /* Analyzer.hop, audio/analysis.py:13; samples per frame, exporter.py:235 */
#define AUR_NBINS      1025   /* fft_size/2+1, audio/analysis.py:10 */
#define AUR_N_BARS       80   /* widgets/rt_widget.py:935 */
#define AUR_N_PARTICLES  50   /* widgets/rt_widget.py:991 */

/* span slots per pixel column (see aur_spans_*) */
#define AUR_K_SPECTRUM    2
#define AUR_K_PARTICLES   2
#define AUR_K_WAVEFORM    5

typedef struct aur_ctx aur_ctx;

/* ---- library / context --------------------------------------------------------------------- */

int         aur_version(void);                 /* ABI version, bumped on incompatible change */
const char* aur_last_error(void);              /* thread-local, never NULL */
int         aur_device_count(int* out_count);
/* PCI address "dddd:bb:dd.f" of a HIP device, so that the host side can place a rank's threads and its pinned frame ring
 * on the NUMA node the GPU hangs off (aurora_amd/sharding.py: bind_near_device).  No counterpart in the reference, which
 * exports in one process on the CPU (export/worker.py). */
int         aur_device_pci_bus_id(int device, char* out, int len);

/* Creates a context on `device`: uploads the Hann window (audio/analysis.py:9) and the FFT twiddle
 * tables (computed in float64, rounded to float32).  Replaces Analyzer.__init__ (analysis.py:6-14). */
int  aur_create(int device, aur_ctx** out);
void aur_destroy(aur_ctx* ctx);

/* Copies the float32[2048] analysis window held by the context to the host (parity tests compare it
 * with scipy.signal.get_window('hann', 2048, fftbins=True).astype(float32)). */
int  aur_get_window(aur_ctx* ctx, float* h_window /*[2048]*/);

/* ---- analysis: replaces _read_frame + Analyzer.compute --------------------------------------- */

/* Fused per-frame analysis for frames [frame_begin, frame_begin + n_frames) of one track.
 *
 * Replaces, per frame f (reference: export/exporter.py:227-238 `_read_frame`, audio/analysis.py:16-33
 * `Analyzer.compute`, widgets/rt_widget.py:824 energy):
 *   samples  = zeros(1024); samples[:n] = mono(audio[pos[f] : pos[f]+hop])[:1024]      (mono = channel mean)
 *   buffer   = [samples(f-1) | samples(f)]   (samples(-1) = 0)      -> * periodic Hann -> rfft(2048) -> |X|
 *   d_logspec[f]   = log1p(|X|)                       float32[1025]   (optional, NULL to skip)
 *   d_flux_sum[f]  = sum(max(0, |X|(f) - |X|(f-1)))   float32  (|X|(-1) = 0; Analyzer.compute divides by 1025)
 *   d_meanabs[f]   = mean(|samples|)                  float32, numpy pairwise-summation order (bit-exact)
 *   d_samples[f]   = samples                          float32[1024]   (optional, NULL to skip)
 *   d_bandmax[f]   = (max(logspec[:80]), max(logspec[:50]))  float32[2] (optional, NULL to skip)
 * Output arrays are indexed by (f - frame_begin).  The sliding-buffer / flux history of frames before
 * frame_begin is rebuilt from the audio itself (time-shard halo), so any frame range gives the same
 * values as a run from frame 0.
 *
 * d_audio  : float32 [n_samples, channels] interleaved (soundfile layout); channels must be 1 or 2 (more: see
 *            aur_mix_channels).
 * d_pos    : int64 [n_track_frames] sample position of EVERY frame of the track (exporter.py:230),
 *            computed on the host in float64; entries frame_begin-2 .. frame_begin+n_frames-1 are read.
 * hop      : max(1, int(sr/fps)) (exporter.py:229).
 */
int aur_stft_frames(aur_ctx* ctx,
                    const float* d_audio, int64_t n_samples, int channels,
                    const int64_t* d_pos, int64_t frame_begin, int n_frames, int hop,
                    float* d_logspec, float* d_flux_sum, float* d_meanabs,
                    float* d_samples, float* d_bandmax, void* stream);

/* Mono mix of a track with any channel count: d_mono[i] = mean(d_audio[i, :]) in float32, evaluated in the order numpy's
 * `y.mean(axis=1)` uses (export/exporter.py:233-234): left to right from 0 below 8 channels, eight strided accumulators
 * joined pairwise from 8 on, divided by the channel count.  aur_stft_frames takes 1 or 2 channels itself (its loads are
 * specialised for them); tracks with more (5.1, 7.1 files) are mixed once per track with this call and analysed as mono,
 * which gives the same samples bit for bit.  d_audio: float32 [n_samples, channels]; d_mono: float32 [n_samples];
 * 1 <= channels <= 64. */
int aur_mix_channels(aur_ctx* ctx, const float* d_audio, int64_t n_samples, int channels, float* d_mono, void* stream);

/* One streaming Analyzer.compute step (audio/analysis.py:25-33) for callers that keep Analyzer.buffer
 * themselves (the realtime caller audio/input.py:276-293 and the drop-in Analyzer.compute):
 *   windowed = buffer * window; mag = |rfft(windowed)|; d_flux_sum[0] = sum(max(0, mag - prev_mag));
 *   d_prev_mag <- mag (in place); d_logspec = log1p(mag).
 * d_buffer: float32[2048] already slid by the caller (analysis.py:18-23); d_prev_mag: float32[1025] in/out. */
int aur_stft_buffer(aur_ctx* ctx, const float* d_buffer, float* d_prev_mag, float* d_logspec,
                    float* d_flux_sum, void* stream);

/* The kernels' log1p (replaces np.log1p at audio/analysis.py:33 for magnitudes x >= 0) applied element-wise:
 * d_y[i] = log1p(d_x[i]).  Exposed so the accuracy of the device routine can be tested over its whole range. */
int aur_log1p_f32(aur_ctx* ctx, const float* d_x, float* d_y, int64_t n, void* stream);

/* ---- smoothing scan over frames --------------------------------------------------------------- */

typedef struct aur_scan_params {
    double sensitivity;      /* RTVisualizerWidget.waveform_sensitivity (rt_widget.py:140-145), >= 0.05 */
    double amp_alpha;        /* _amp_alpha = 0.2 (rt_widget.py:57, :487) */
    double grad_min;         /* _grad_min = 0.0 (rt_widget.py:54) */
    double grad_max;         /* _grad_max = 1.0 (rt_widget.py:55) */
    int    grad_curve;       /* 0 linear, 1 ease-in, 2 ease-out, 3 smoothstep (rt_widget.py:490-495) */
    uint8_t grad_a[4];       /* (32,255,255) rt_widget.py:51 */
    uint8_t grad_b[4];       /* (255,0,255)  rt_widget.py:52 */
    double flux_alpha;       /* Analyzer.flux_alpha = 0.9 (analysis.py:12) */
    /* carried state at the first frame of this call (0 at track start); updated on return path via d_state_out */
    double amp_ema0;
    double flux_smooth0;
    double phase0;
} aur_scan_params;

/* Sequential-order (bit-exact, float64) recurrences over n_frames consecutive frames:
 *   energy[f]      = float(clip(float32(meanabs[f]) * sensitivity, 0, 1))              rt_widget.py:824
 *   amp_ema        = amp_alpha*amp_ema + (1-amp_alpha)*energy[f]  -> pen_rgb[f]         rt_widget.py:485-499
 *   flux_smooth[f] = flux_alpha*flux_smooth + (1-flux_alpha)*(flux_sum[f]/1025)         analysis.py:29-30
 *   phase[f]       = (phase + 0.04*(0.5+energy[f])) % (2*pi)                            rt_widget.py:825
 * d_pen_rgb: uint8 [n_frames,4] (r,g,b,0).  d_state_out: float64[3] = (amp_ema, flux_smooth, phase) after the
 * last frame (carried smoothing state for the next time shard); may be NULL.  Any output may be NULL.
 */
int aur_scan_frames(aur_ctx* ctx, const float* d_meanabs, const float* d_flux_sum, int n_frames,
                    const aur_scan_params* params,
                    double* d_energy, uint8_t* d_pen_rgb, double* d_flux_smooth, double* d_phase,
                    double* d_state_out, void* stream);

/* ---- rasteriser -------------------------------------------------------------------------------- */

/* Column spans: uint32 [n_frames, K, W]; element = r0 | (r1 << 16), a half-open lit row interval
 * [r0, r1) of that pixel column (0 = empty).  64 <= W <= 16384 (any integer: the reference UI offers 64..3840),
 * 16 <= H <= 16384.                                                                                 */

/* `_draw_spectrum` linear branch (rt_widget.py:934-940, :982-988): 80 vertical width-2 lines. K = 2. */
int aur_spans_spectrum_linear(aur_ctx* ctx, const float* d_logspec /*[F,1025]*/, const double* d_energy /*[F]*/,
                              int n_frames, int W, int H, uint32_t* d_spans, void* stream);

/* `_draw_particles` (rt_widget.py:990-1000): 50 width-2 points. K = 2. */
int aur_spans_particles(aur_ctx* ctx, const float* d_logspec, int n_frames, int W, int H,
                        uint32_t* d_spans, void* stream);

/* `_draw_waveform` linear branch (rt_widget.py:903-913, :932): width-2 SquareCap/BevelJoin polyline through
 * (x, mid - samples[col2samp[x]]*amp).  d_col2samp: int32[W] = int((x/max(1,W-1))*(1023)) computed on the host
 * in float64 (rt_widget.py:908).  K = 5. */
int aur_spans_waveform_linear(aur_ctx* ctx, const float* d_samples /*[F,1024]*/, const int32_t* d_col2samp,
                              const double* d_energy, int n_frames, int W, int H, uint32_t* d_spans, void* stream);

/* Fills n_frames RGB24 frames (uint8 [F,H,W,3], tightly packed = `_qimage_to_numpy` output, exporter.py:84-95,
 * i.e. the rawvideo rgb24 wire format of build_ffmpeg_cmd, exporter.py:429-434): background (13,13,18)
 * (rt_widget.py:829), pen colour d_pen_rgb[f] on every pixel inside a span.                           */
int aur_raster_spans(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb /*[F,4]*/,
                     int n_frames, int W, int H, uint8_t* d_frames, void* stream);

/* The same frames through the linear fill (one 4 KB-aligned 4 KB piece of the batch per workgroup, one store per lane:
 * the store shape that reaches the HBM write ceiling of MI355X).  d_work: device work area of
 * aur_raster_work_bytes(n_frames, W) bytes, 16-byte aligned: per-frame column summaries the fill looks up instead of
 * re-reading spans.  aur_span_summary builds them from d_spans; aur_raster_spans_ws does that itself unless
 * work_ready != 0 (d_work already holds the summaries of exactly these spans).  aur_raster_spans_ws falls back to
 * aur_raster_spans when d_work is NULL or the shape does not suit (W not a multiple of 16, fewer than 256 16-byte
 * columns per frame, batch above 2^31 columns).  Replaces the same reference code as aur_raster_spans; results are
 * byte-identical to it. */
int aur_raster_work_bytes(int n_frames, int W, size_t* out_bytes);
int aur_span_summary(aur_ctx* ctx, const uint32_t* d_spans, int K, int n_frames, int W, int H, void* d_work,
                     size_t work_bytes, void* stream);
int aur_raster_spans_ws(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb /*[F,4]*/,
                        int n_frames, int W, int H, uint8_t* d_frames, void* d_work, size_t work_bytes, int work_ready,
                        void* stream);

/* aur_raster_spans_ws without a work-area argument and without any allocation.  No counterpart in the reference: it is for
 * callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  The work area is a file-scope device array of AUR_STATIC_WORK_BYTES inside the library (one copy per GPU,
 * held until the process ends); a batch whose aur_raster_work_bytes(n_frames, W) exceeds it is rejected with AUR_EINVAL
 * (3840 wide: at most 148 frames per call; 1920: 297; 1280: 446).  Shapes the linear fill does not take (see
 * aur_raster_spans_ws) go to aur_raster_spans, which needs no work area.  All calls on one GPU share that GPU's array:
 * calls on one stream are ordered by the stream and safe; a call that arrives on ANOTHER stream while the previous call on
 * that GPU has not finished would race on the array and is refused with AUR_EINVAL (the library keeps, per GPU, the last
 * call's stream and an event recorded behind it, created in aur_create and looked at with hipEventQuery: no allocation,
 * no synchronisation).  Order the streams, or give each stream a work area of its own (aur_raster_spans_ws).  Streams
 * under graph capture are not tracked: launches of one captured graph are ordered by the graph's own edges.
 * Results are byte-identical to aur_raster_spans_ws / aur_raster_spans. */
#define AUR_STATIC_WORK_BYTES (3 * 1024 * 1024)
int aur_raster_spans_static(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb /*[F,4]*/,
                            int n_frames, int W, int H, uint8_t* d_frames, void* stream);

/* ---- "Spectrum - Radial": radial path, centre-disc envelope, bit planes ------------------------------------ */

/* Stage 1 of the radial spectrum (rt_widget.py:938-951): spectrum[:80] / max (float32) -> optional mirror
 * (np.concatenate([spec, spec[::-1][1:-1]])) -> _smooth_closed(radial_wave_smoothness) (rt_widget.py:1121-1137).
 * d_vals1: float64 [n_frames, L1], L1 = (mirror ? 158 : 80) * (1 + wave_smoothness / 25)  (x1 if smoothness == 0). */
int aur_radial_stage1(aur_ctx* ctx, const float* d_logspec /*[F,1025]*/, int n_frames, int mirror,
                      int wave_smoothness, double* d_vals1, int L1, void* stream);

/* The _radial_prev temporal EMA (rt_widget.py:953-958) over consecutive frames, in place on d_vals1:
 *   prev = alpha * prev + (1 - alpha) * cur;  the first frame copies when has_state == 0 (prev is None).
 * d_state: float64 [L1], read when has_state != 0 (carried state of the previous time shard), always written with
 * the state after the last frame.  Skip the call when alpha == 0 (the reference skips the block). */
int aur_radial_ema(aur_ctx* ctx, double* d_vals1, int n_frames, int L1, double alpha, int has_state, double* d_state,
                   void* stream);

/* Stage 2 (rt_widget.py:960-974): _smooth_closed(radial_smooth_amount) -> L2 + 1 vertices per frame,
 *   r = inner + span * v, (x, y) = (W/2 + cos(a_i) * r, H/2 + sin(a_i) * r), i = 0..L2, v = spec_draw[min(i, L2-1)].
 * d_cos_sin: float64 [L2 + 1, 2] = (cos a_i, sin a_i), a_i = 2 pi (i / L2) - pi/2 + deg2rad(rotation), computed on the
 * host with the reference's expression (rt_widget.py:966).  d_verts: float64 [n_frames, L2 + 1, 2].
 * L2 = L1 * (1 + smooth_amount / 25)  (x1 if smooth_amount == 0 or L1 <= 3). */
int aur_radial_vertices(aur_ctx* ctx, const double* d_vals1, const double* d_energy, int n_frames, int L1,
                        int smooth_amount, const double* d_cos_sin, int W, int H, double* d_verts, int L2,
                        void* stream);

typedef struct aur_envelope_params {
    int    radial_mirror;          /* rt_widget.py:1026 */
    int    edge_waviness;          /* 0..100, rt_widget.py:1020 */
    int    feather_audio_enabled;  /* rt_widget.py:1021 */
    int    feather_audio_amount;   /* 0..100 */
    double feather_sensitivity;    /* rt_widget.py:60, :1030 */
    double waveform_sensitivity;   /* rt_widget.py:59, :1030 */
    double rotation_rad;           /* np.deg2rad(radial_rotation_deg), rt_widget.py:1035 */
} aur_envelope_params;

/* Envelope of the centre disc, `_draw_center_image_and_mask` (rt_widget.py:1007-1051): 361 vertices per frame,
 *   r = inner + amp_noise * fbm(2 a, phase) + amp_audio * spec_norm[si], soft-limited by tanh.
 * d_cos_sin360: float64 [361, 2] host table of (cos a_i, sin a_i), a_i = 2 pi (i / 360) - pi/2 + rotation.
 * d_verts: float64 [n_frames, 361, 2]. */
int aur_envelope_vertices(aur_ctx* ctx, const float* d_logspec, const double* d_energy, const double* d_phase,
                          int n_frames, const aur_envelope_params* params, const double* d_cos_sin360, int W, int H,
                          double* d_verts, void* stream);

/* Feather rings (`feather_enabled`, rt_widget.py:1098-1117): six scaled copies k = 1..6 of
 *   r = (inner + amp_noise sin(2 a_i + phase) + amp_audio spec_norm[si]) (1 + 0.015 k), 361 vertices each.
 * d_rings: float64 [n_frames, 6, 361, 2].  Rasterise them with aur_mask_polygon (6 n_frames polygons) into planes
 * [n_frames, 6, H, ceil(W/32)] and apply them with aur_blend_planes.
 *
 * aur_blend_planes: source-over of K <= 8 constant colours h_rgba[K][4] (host array, r g b a), plane k wherever its bit
 * is set, in plane order, in place on finished RGB24 frames: `p.fillPath(feather_path, QColor(13, 13, 18,
 * max(10, 70 - 10 k)))`, k = 1..6 (rt_widget.py:1099-1117).  Qt's premultiplied ARGB32 arithmetic,
 *   d' = div255(c a) + div255(d (255 - a)), div255(x) = (x + (x >> 8) + 0x80) >> 8; a = 255 overwrites
 * (parity-unpinned against Qt's raster engine).  d_planes: uint32 [n_frames, K, H, ceil(W/32)]. */
int aur_feather_ring_vertices(aur_ctx* ctx, const float* d_logspec, const double* d_energy, const double* d_phase,
                              int n_frames, const aur_envelope_params* params, const double* d_cos_sin360, int W, int H,
                              double* d_rings, void* stream);
int aur_blend_planes(aur_ctx* ctx, const uint32_t* d_planes, int K, const uint8_t* h_rgba /*[K][4]*/, int n_frames,
                     int W, int H, uint8_t* d_frames, void* stream);
/* The same blend from planes that cover only a box of the frame: d_planes uint32 [n_frames, K, box_h, ceil(box_w/32)]
 * holds pixel columns box_x .. box_x + box_w (box_x a multiple of 32) of rows box_y .. box_y + box_h.  The disc and its
 * rings stay within 0.34 min(W, H) of the centre, a quarter of a 1080p frame: planes of the box are a quarter of the
 * plane traffic.  Same pixels as aur_blend_planes on full planes, bit for bit. */
int aur_blend_planes_box(aur_ctx* ctx, const uint32_t* d_planes, int K, const uint8_t* h_rgba /*[K][4]*/, int n_frames,
                         int W, int H, int box_x, int box_y, int box_w, int box_h, uint8_t* d_frames, void* stream);

/* "Waveform - Circular" (`_draw_waveform`, circular branch, rt_widget.py:871-896): the closed path through all
 * 1024 samples of a frame (segs = min(720, n), step = max(1, int(n / segs)) = 1),
 *   r = 0.28 min(W,H) + float(samples[i]) * 0.22 min(W,H) (0.5 + 0.6 energy),  (x, y) = (W/2 + cos(a_i) r, H/2 + sin(a_i) r).
 * d_samples: float32 [n_frames, 1024] (aur_stft_frames' d_samples); d_cos_sin: float64 [1024, 2] host table of
 * (cos a_i, sin a_i), a_i = 2 pi (i / 1023) - pi/2 + deg2rad(rotation) (rt_widget.py:887).
 * d_verts: float64 [n_frames, 1024, 2]; stroke it with aur_mask_stroke_closed (`p.drawPath(path)`, :902). */
int aur_wave_circular_vertices(aur_ctx* ctx, const float* d_samples, const double* d_energy, int n_frames,
                               const double* d_cos_sin, int W, int H, double* d_verts, void* stream);

/* Bit planes: uint32 [n_frames, H, ceil(W/32)], bit (x & 31) of word (x >> 5) = pixel (x, row) is covered.
 * aur_mask_stroke_closed: width-2 BevelJoin stroke of the closed polygon through d_verts [F, n_vertices, 2]
 *   (`p.drawPath(path)` with the width-2 pen, rt_widget.py:975-981);
 * aur_mask_polygon: even-odd fill of the implicitly closed polygon, sampled at pixel centres
 *   (`p.fillPath(path, QColor(30,30,40))` under the clip, rt_widget.py:1056-1096; edge antialiasing not reproduced);
 *   any number of edge crossings per pixel row (every crossing toggles one bit of the row, the fill is the prefix XOR:
 *   deterministic whatever the polygon, e.g. the noisy fill polygon of the circular waveform, rt_widget.py:856-864).
 * Both overwrite the whole plane. */
int aur_mask_stroke_closed(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H,
                           uint32_t* d_plane, void* stream);
int aur_mask_polygon(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H,
                     uint32_t* d_plane, void* stream);
/* aur_mask_polygon into planes of a box of the frame (layout as aur_blend_planes_box): crossings are computed in frame
 * coordinates, so the bits are those of the full-frame plane inside the box; what lies outside the box is not recorded. */
int aur_mask_polygon_box(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H, int box_x,
                         int box_y, int box_w, int box_h, uint32_t* d_plane, void* stream);
/* Stroke of half width `half_width` (pen width / 2) of the path translated by (dx, dy), closed or open:
 * `_draw_shadow_path` (rt_widget.py:579-594) strokes a path with a QPen of width max(2, shadow_blur), default
 * SquareCap / BevelJoin, after `p.translate(dist cos(angle), dist sin(angle))`; the radial / circular paths end with
 * closeSubpath, the envelope path of the centre disc does not (closed = 0: no closing segment, no join at the ends,
 * square caps).  half_width = 1, dx = dy = 0, closed = 1 is aur_mask_stroke_closed bit for bit.
 * round_join != 0: RoundCap / RoundJoin pen = every pixel centre within half_width of the polyline (segment rectangles
 * and a disc at every vertex): the shadow of "Waveform - Linear" (rt_widget.py:915-928). */
int aur_mask_stroke_path(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H,
                         double half_width, double dx, double dy, int closed, int round_join, uint32_t* d_plane,
                         void* stream);

/* The polyline of "Waveform - Linear" as vertices (rt_widget.py:904-913): d_verts float64 [n_frames, W, 2] =
 * (x, float(mid - samples[col2samp[x]] * amp)), the values aur_spans_waveform_linear rasterises.  Needed when the
 * shadow of the mode is drawn (the path is then stroked through bit planes).  W <= 4097. */
int aur_wave_linear_vertices(aur_ctx* ctx, const float* d_samples, const int32_t* d_col2samp, const double* d_energy,
                             int n_frames, int W, int H, double* d_verts, void* stream);

/* d_plane &= ~(d_excl_a | d_excl_b) per frame (either may be NULL; excl_b_static != 0: one plane for all frames), and the
 * plane of frame f is cleared when d_enable != NULL and d_enable[f] == 0.  Used for `_apply_fill` (rt_widget.py:856-864,
 * called at :898 and :977 before the path is stroked and before the centre disc): the fill polygon of the closed path
 * only shows where neither the stroke nor the disc paints later, and only in frames with energy >= threshold; its
 * colour over the plain background is a per-export constant, applied with aur_blend_planes. */
int aur_plane_exclude(aur_ctx* ctx, uint32_t* d_plane, const uint32_t* d_excl_a, const uint32_t* d_excl_b,
                      int excl_b_static, const uint8_t* d_enable /*[F]*/, int n_frames, int W, int H, void* stream);

/* Glow decoration, `_draw_glow_path` -> `_draw_glow_offscreen` (rt_widget.py:598-616, :634-653; export sets
 * _offscreen_paint, exporter.py:118): before the closed path of "Spectrum - Radial" / "Waveform - Circular" is stroked,
 * it is stroked `layers` times with RoundCap / RoundJoin pens of width max(3, int(2 + radius (0.35 + 0.9 t))) in the glow
 * colour with alpha int(255 base_alpha intensity 0.55 (1 - t)^2), CompositionMode_Plus.  Layer planes come from
 * aur_mask_stroke_path(round_join = 1).  aur_add_planes adds, for frame f and layer k, the premultiplied colour
 * d_pm_rgb[f][k] = (div255(r a), div255(g a), div255(b a), 0) (computed by the host side from the frame's energy) to every
 * pixel of plane k that is in neither exclusion plane (d_excl_a [n_frames, H, W32]; d_excl_b per frame, or one plane for
 * all frames when excl_b_static != 0; either may be NULL: the stroke and the disc, which the reference paints afterwards
 * with opaque colours), saturating each channel at 255 (comp_func_Plus on an opaque destination; the reference's
 * antialiased edge coverage is not reproduced).
 * d_planes: uint32 [K, n_frames, H, ceil(W/32)] (layer-major), K <= 16; d_pm_rgb: uint8 [n_frames, K, 4], 4-byte aligned. */
int aur_add_planes(aur_ctx* ctx, const uint32_t* d_planes, int K, const uint8_t* d_pm_rgb, const uint32_t* d_excl_a,
                   const uint32_t* d_excl_b, int excl_b_static, int n_frames, int W, int H, uint8_t* d_frames,
                   void* stream);

/* CompositionMode_Multiply of the constant straight-alpha colour h_rgba[4] (HOST) on the pixels of d_plane uint32
 * [n_frames, H, ceil(W/32)], in place: `_apply_fill` with radial_fill_blend = "multiply" (rt_widget.py:859-862) when the
 * frame under it is a background image (over the plain background the result is one colour, composed by the caller).
 * Qt's multiply_op with an opaque destination: d' = div255(d (div255(c a) + 255 - a)) per channel. */
int aur_multiply_plane(aur_ctx* ctx, const uint32_t* d_plane, const uint8_t* h_rgba, int n_frames, int W, int H,
                       uint8_t* d_frames, void* stream);

/* RGB24 frames of "Spectrum - Radial" / "Waveform - Circular" (paint_frame, rt_widget.py:821-852): background
 * (13,13,18), stroke plane in the pen colour, disc plane in (30,30,40) on top (an all-zero static plane when the mode
 * has no disc).  Widths that are not a multiple of 16 take a per-pixel kernel.  disc_static != 0: d_disc_plane holds ONE
 * plane [H, ceil(W/32)] used for every frame (the envelope does not depend on the frame when edge_waviness and the
 * audio term are off, which is the export default). */
int aur_raster_planes(aur_ctx* ctx, const uint32_t* d_stroke_plane, const uint32_t* d_disc_plane, int disc_static,
                      const uint8_t* d_pen_rgb /*[F,4]*/, int n_frames, int W, int H, uint8_t* d_frames, void* stream);

/* Glow decoration of a closed path (`_draw_glow_path` -> `_draw_glow_offscreen`, rt_widget.py:598-616, :634-653): K
 * antialiased RoundCap / RoundJoin strokes of the same path with half pen widths h_half_widths[K] (HOST array, paint
 * order), CompositionMode_Plus in the per-frame premultiplied layer colours d_pm_rgb uint8 [F][K][4], in place on
 * d_frames, wherever neither d_excl_a [F][H][ceil(W/32)] nor d_excl_b ([F] or one static plane) has a bit (what the
 * reference paints afterwards with opaque colours; either may be NULL).  One distance pass serves all layers: coverage
 * of a layer = area of the pixel within half its pen width of the path, rim taken as straight across the pixel.
 * d_verts float64 [F][n_vertices][2], n_vertices <= 4096; d_work: aur_glow_work_bytes(...) bytes, 256-byte aligned. */
int aur_glow_work_bytes(int n_frames, int n_vertices, size_t* out_bytes);
int aur_glow_path(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H, int K,
                  const double* h_half_widths, const uint8_t* d_pm_rgb, const uint32_t* d_excl_a,
                  const uint32_t* d_excl_b, int excl_b_static, void* d_work, size_t work_bytes, uint8_t* d_frames,
                  void* stream);

/* ---- antialiased centre disc, centre image, background image ---------------------------------- */

/* What fills the centre disc of "Spectrum - Radial" (widgets/rt_widget.py:1060-1094).
 * mode 0: `p.fillPath(path, QColor(30, 30, 40))` under `setClipPath(path)`: the constant premultiplied colour, coverage of
 *         the fill multiplied with the coverage of the clip (Qt intersects the two span lists);
 * mode 1: `p.drawImage(dx, dy, img.scaled(out_w, out_h, IgnoreAspectRatio, SmoothTransformation))` under the clip, scaled
 *         per frame from d_image (premultiplied 0xAARRGGBB [img_h][img_w]); d_placement int32 [F or 1][4] = out_w, out_h,
 *         dx, dy (aur_center_image_placement);
 * mode 2: the same with d_image already scaled to placement's out_w x out_h (aur_image_scale; static placement). */
typedef struct aur_disc_source {
    int32_t mode;
    uint32_t colour_argb_pm;
    const uint32_t* d_image;
    int32_t img_w, img_h;
    const int32_t* d_placement;
    int32_t placement_static;
} aur_disc_source;

/* QImage::scaled(dst_w, dst_h, Qt::IgnoreAspectRatio, Qt::SmoothTransformation) of a premultiplied ARGB32 image
 * (rt_widget.py:567, :1090; Qt's qimagescale.cpp: 8-bit bilinear weights when an axis grows, 14-bit area average when it
 * shrinks).  d_src uint32 [src_h][src_w], d_dst uint32 [dst_h][dst_w]. */
int aur_image_scale(aur_ctx* ctx, const uint32_t* d_src, int src_w, int src_h, uint32_t* d_dst, int dst_w, int dst_h,
                    void* stream);

/* Placement of the centre image per frame (rt_widget.py:1065-1088): bounding rect of the clip path adjusted by 2 px,
 * cover scale, `center_image_zoom`, `center_motion` zoom and jitter from the frame's energy and phase.
 * d_verts float64 [F or 1 (verts_static)][n_vertices][2]; d_placement int32 [F][4] = out_w, out_h, dx, dy. */
int aur_center_image_placement(aur_ctx* ctx, const double* d_verts, int n_vertices, int verts_static,
                               const double* d_energy, const double* d_phase, int n_frames, int img_w, int img_h,
                               int zoom_percent, int center_motion, int32_t* d_placement, void* stream);

/* The centre disc with the reference's antialiasing (rt_widget.py:1056-1096): exact-area even-odd coverage of the
 * envelope polygon per pixel (26.6 vertices, 1/256-pixel cells, 0..255) and Qt's premultiplied source-over with that
 * coverage, in place on d_frames uint8 [F][H][W][3].  The polygon must lie inside the box (box_x, box_y, box_w, box_h).
 * d_full_plane (optional) receives the bit plane uint32 [F][H][ceil(W/32)] of the pixels with full coverage (only the
 * words of the box are written: clear it first); d_frames may be NULL when only that plane is wanted.  skip_full: in
 * mode 0 the pixels of full coverage are left alone (the plane fill painted them already). */
int aur_disc_composite(aur_ctx* ctx, const double* d_verts, int n_vertices, int verts_static, int n_frames, int W, int H,
                       int box_x, int box_y, int box_w, int box_h, const aur_disc_source* src, int skip_full,
                       uint32_t* d_full_plane, uint8_t* d_frames, void* stream);

/* The same source-over for a frame-independent disc (the export default: no waviness, no audio term): its partially
 * covered pixels as a list -- d_pixel_index uint32 [n] = y * W + x (every index < W * H: the caller's contract),
 * d_coverage uint8 [n] = effective coverage -- blended in colour_argb_pm (premultiplied) into every frame. */
int aur_blend_pixels(aur_ctx* ctx, const uint32_t* d_pixel_index, const uint8_t* d_coverage, int n_pixels,
                     uint32_t colour_argb_pm, int n_frames, int W, int H, uint8_t* d_frames, void* stream);

/* The frame every export frame starts from when a background image is set (`paint_frame` rt_widget.py:829-830,
 * `_draw_bg_image` :519-545): fillRect (13,13,18), the (already scaled) premultiplied image drawn at (x, y) -- or tiled
 * from (x, y) in steps of its size over the frame's own tiling range -- then fillRect(0, 0, w, h, QColor(0, 0, 0,
 * dim_alpha)).  d_base uint8 [H][W][3]. */
int aur_compose_background(aur_ctx* ctx, const uint32_t* d_image, int img_w, int img_h, int tile, int x, int y,
                           int dim_alpha, int W, int H, uint8_t* d_base, void* stream);

/* Sets (d_base != NULL) or clears (NULL) the base frame uint8 [H][W][3] that aur_raster_spans*, aur_raster_planes and
 * their fallbacks paint on instead of the plain (13,13,18) background.  The pointer must stay valid until it is cleared
 * or the context is destroyed; W, H must match the frames rendered afterwards. */
int aur_set_background(aur_ctx* ctx, const uint8_t* d_base, int W, int H);

/* A batch of closed-path frames in one call and without a stroke plane, for renders that need none afterwards (no fill /
 * shadow / glow decoration): pass 1 of aur_raster_planes (background or base frame, disc rows), then the width-2 BevelJoin
 * stroke of d_verts painted straight into the frames in the pen colour, leaving out what the disc covers.  The same frames
 * as aur_mask_stroke_closed + aur_raster_planes, bit for bit, without the plane clear, the stroke plane and the patch pass.
 * Widths that are multiples of 32 and 16-byte aligned batches only: AUR_EUNSUPPORTED otherwise (use the two calls).
 * d_disc_rows: NULL, or the row table of d_disc_plane from aur_disc_rows (uint32 [1 or F][H]): a caller whose disc does
 * not change from batch to batch -- the export default -- builds it once instead of once per call. */
int aur_raster_closed_path(aur_ctx* ctx, const double* d_verts, int n_vertices, int n_frames, int W, int H,
                           const uint32_t* d_disc_plane, int disc_static, const uint32_t* d_disc_rows,
                           const uint8_t* d_pen_rgb, uint8_t* d_frames, void* stream);

/* Row table of disc planes uint32 [n_planes][H][W32]: per row the pixel interval x0 | x1 << 16 the disc covers, 0 for an
 * empty row, 0xffffffff for a row whose disc is not one interval (its pixels are then painted from the plane itself).
 * What pass 1 of the closed-path fill reads instead of the plane (4 bytes per row). */
int aur_disc_rows(aur_ctx* ctx, const uint32_t* d_disc_plane, int n_planes, int W, int H, uint32_t* d_rows, void* stream);

/* ---- wire format ------------------------------------------------------------------------------- */

/* Opt-in planar yuv420p wire format: the reference pipes rgb24 into FFmpeg
 * (export/exporter.py:429-434) and FFmpeg's swscale converts to yuv420p (:485-490); converting on the GPU halves the
 * bytes that cross PCIe and the pipe (1.5 instead of 3 per pixel).  BT.601 limited range, 8-bit integer form, chroma
 * from the rounded 2x2 mean (csrc/wire.hip states the formula; swscale's own rounding is parity-unpinned).
 * d_rgb: [F, H, W, 3]; d_yuv: [F, W*H*3/2] = Y [H, W], Cb [H/2, W/2], Cr [H/2, W/2] per frame.
 * W must be a multiple of 4 and H even. */
int aur_rgb24_to_yuv420p(aur_ctx* ctx, const uint8_t* d_rgb, int n_frames, int W, int H, uint8_t* d_yuv, void* stream);

/* The span modes straight into the yuv420p wire format: the frame aur_raster_spans would produce, converted as
 * aur_rgb24_to_yuv420p converts it, without materialising it (a span-mode frame holds two colours, so its planes follow
 * from the spans: 1.5 B per pixel written once instead of 3 written + 3 read + 1.5 written).  d_spans, K, d_pen_rgb as
 * aur_raster_spans (d_spans 16-byte aligned); d_yuv uint8 [F][H * W * 3 / 2].  Not with a background image set. */
int aur_raster_spans_yuv420p(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb, int n_frames, int W,
                             int H, uint8_t* d_yuv, void* stream);

/* The same two conversions into nv12, the semi-planar layout FFmpeg's VAAPI encoders take (the reference asks FFmpeg for
 * "-vf format=nv12,hwupload ... -pix_fmt nv12" on that path, export/exporter.py:459-466, and leaves the rgb24 -> nv12
 * conversion to swscale): Y [H, W], then one plane [H/2, W] of interleaved Cb Cr pairs.  Same arithmetic, same
 * constraints, same size (W*H*3/2 per frame) as the yuv420p entries; opt-in, view_state["hip_wire_format"] = "nv12". */
int aur_rgb24_to_nv12(aur_ctx* ctx, const uint8_t* d_rgb, int n_frames, int W, int H, uint8_t* d_nv12, void* stream);
int aur_raster_spans_nv12(aur_ctx* ctx, const uint32_t* d_spans, int K, const uint8_t* d_pen_rgb, int n_frames, int W,
                          int H, uint8_t* d_nv12, void* stream);

/* ---- readback ---------------------------------------------------------------------------------- */

/* Pinned host memory for the frame ring that feeds the FFmpeg pipe (replaces the arr.copy()/tobytes()
 * chain of exporter.py:95, :540-542). */
int aur_host_alloc(void** out, size_t bytes);
int aur_host_free(void* p);
/* hipMemcpyAsync device->host on `stream`. */
int aur_readback_async(aur_ctx* ctx, const void* d_src, size_t bytes, void* h_dst, void* stream);
int aur_stream_synchronize(aur_ctx* ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AURORA_HIP_H */
