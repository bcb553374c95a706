// Fused per-frame analysis kernel for gfx950 (MI355X): gather -> mono mix -> zero-pad -> Hann ->
// 2048-point real FFT -> |X| -> log1p / spectral-flux partial / mean|x| / band maxima.
//
// Replaces export/exporter.py:227-238 (_read_frame) + audio/analysis.py:16-33 (Analyzer.compute) +
// the energy of widgets/rt_widget.py:824 for a whole range of frames per launch.
//
// Arithmetic: the reference's np.fft.rfft(float32) resolves to numpy's float64 pocketfft loop ('dd->D' is the
// first matching loop of the ufunc) and the result is rounded to complex64, so the reference spectrum is a
// float64 FFT of the float32 product buffer*window.  Leakage bins sit 8-9 decades below the peak of a frame,
// which a float32 butterfly network cannot resolve (its rounding error is relative to the peak).  The FFT here
// therefore runs in float64 -- one v_fma_f64 per 4 cycles and wave on the MI355X vector ALU, the rate of unpacked float32 --
// and only the
// window product (as in the reference), |X| -> log1p and the flux run in float32.
//
// Design (one 64-lane wavefront per frame, frames_per_wave consecutive frames per wave):
//  * the 2048-point real FFT is a 1024-point complex FFT of z[n] = x[2n] + i x[2n+1] plus an unpack pass;
//  * 1024 = 16 x 16 x 4: lane l holds z[l + 64 j] (16 complex values in registers).  Pass 1 is a radix-16
//    DFT in registers, pass 2 a radix-16 DFT after an LDS transpose, pass 3 a radix-4 after a second LDS
//    transpose.  The index split is chosen so pass 3 leaves Z[l + 64 j] in lane l again, which is the
//    coalesced store order of the spectrum (bins l + 64 j) -- no third transpose;
//  * the transposes move real and imaginary parts separately (ds_write_b64 / ds_read_b64, rows padded to 66
//    doubles: conflict-free both ways), so a wave needs 8.25 KB of LDS and 12 waves fit a CU next to the tables;
//  * the second transpose hands every lane its radix-4 groups in partner pairs: next to the groups that produce
//    Z[l + 64 j] (j in {0,1,4,5,8,9,12,13}) a lane reads the groups that produce Z[1024 - l - 64 j], so the real-FFT
//    unpack finds both operands of a bin pair in its own registers (no cross-lane traffic after pass 3);
//  * wave reductions (flux, mean|x|, band maxima) are DPP row operations, not LDS round trips;
//  * the tables (window, W1024^(l k1), W64^(r q), the log table) are staged in LDS once per workgroup; the lane's 16 window
//    pairs, its twelve W64^(r q) factors (applied behind the second transpose) and its unpack twiddles 0.5 W2048^l W32^j
//    then live in registers across the frame loop, W1024^(l k1) is read from LDS per frame (one ds_read_b128 each);
//  * the constant twiddles inside the 16-point DFTs are folded into the additions that consume them (radix4_tw1/2/3);
//  * a wave walks consecutive frames, so the previous 1024-sample block (Analyzer.buffer[:1024]) and the
//    previous magnitudes (Analyzer.prev_mag) stay in registers; HBM sees every audio sample once per chunk
//    (plus a 2-block halo per chunk) and every output once;
//  * mean|x| is summed in numpy's pairwise order so the energy is bit-identical to the reference.
//
// Contraction is disabled for the whole library (-ffp-contract=off); fma is used explicitly where wanted.
#include "common.h"

// Spectrum stores: plain by default (-DAUR_STFT_NT=1 builds the non-temporal variant for A/B runs: aux bit 1 = slc).
#ifndef AUR_STFT_NT
#define AUR_STFT_NT 0
#endif
__device__ __forceinline__ void aur_buf_st(float v, const __amdgpu_buffer_rsrc_t rs, int byte_offset) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, byte_offset, 0, AUR_STFT_NT ? 2 : 0);
}

#include <type_traits>

namespace {

#ifndef AUR_STFT_WAVES_PER_SIMD
#define AUR_STFT_WAVES_PER_SIMD 2      // register budget of the frame kernel: 512 / this VGPRs per lane
#endif
#ifndef AUR_STFT_WAVES_PER_WG
#define AUR_STFT_WAVES_PER_WG 4
#endif
#ifndef AUR_STFT_WIDE_STORES
#define AUR_STFT_WIDE_STORES 0         // 1: spectrum rows leave through an LDS stage as dwordx4 stores -- measured SLOWER (below)
#endif
#ifndef AUR_STFT_WIN_REGS
#define AUR_STFT_WIN_REGS 1            // 1: this lane's 16 window pairs live in registers across the frame loop (32 VGPRs)
#endif
#ifndef AUR_STFT_TW3_REGS
#define AUR_STFT_TW3_REGS 1            // 1: the twelve W64^(r q) twiddles of pass 3 live in registers (48 VGPRs).  With the
                                       // window in registers too the kernel fills its 256 VGPRs (two waves per SIMD, no scratch)
                                       // and reads the LDS 31 times less per frame (profiles/r03_stft_lesions.txt)
#endif
#ifndef AUR_STFT_TH_REGS
#define AUR_STFT_TH_REGS 1             // 1: the two unpack twiddles that take a full product (0.5 W2048^lane W32^1, W32^5) are
                                       // loop invariants in registers (8 VGPRs) instead of 8 float64 instructions per frame
#endif
#ifndef AUR_STFT_TW1_REGS
#define AUR_STFT_TW1_REGS 0            // N: the first N of the 15 W1024^(l k1) twiddles of pass 1 in registers (experiment)
#endif
#ifndef AUR_STFT_MID_AT
#define AUR_STFT_MID_AT 2              // unpack slot before which the next block's loads are issued: half of v[] is dead there,
                                       // so the 32 load registers cost no extra pressure (0: before the unpack)
#endif
#ifndef AUR_STFT_WAVES_PER_CU
#define AUR_STFT_WAVES_PER_CU (4 * AUR_STFT_WAVES_PER_SIMD)   // resident waves per CU (limited by LDS when it is not this)
#endif
constexpr int kWavesPerWG = AUR_STFT_WAVES_PER_WG;
constexpr int kXbufDoubles = 16 * 66;            // exchange buffer per wave: 16 rows of 66 doubles, 8448 B
constexpr int kLdsWinBytes = AUR_FFT_SIZE * 4;   // float32 window
constexpr int kTw2Stride = 17;                   // complex doubles per W64 row (16 + 1 pad: rows on distinct banks)
constexpr int kLdsTw2Bytes = 4 * kTw2Stride * 16;   // W64^(r q), complex double
constexpr int kLdsTw1Bytes = 15 * 64 * 16;          // W1024^(l k1), complex double [k1 - 1][l]
constexpr int kLdsLgBytes = ((AUR_LG_ENTRIES * 16 + 63) / 64) * 64;   // log table of log1p_cr: (1 / c_i, log c_i)
constexpr int kLdsTabBytes = kLdsWinBytes + kLdsTw2Bytes + kLdsTw1Bytes + kLdsLgBytes;
#ifndef AUR_STFT_PMAG_LDS
#define AUR_STFT_PMAG_LDS 0
#endif
constexpr int kPmagBytes = AUR_STFT_PMAG_LDS ? 17 * 64 * 4 : 0;      // per wave
constexpr int kLdsBytes = kLdsTabBytes + kWavesPerWG * (kXbufDoubles * 8 + kPmagBytes);

struct cd {
    double x, y;
};

__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd w) {
    return {fma(a.x, w.x, -(a.y * w.y)), fma(a.x, w.y, a.y * w.x)};
}
__device__ __forceinline__ cd mul_mi(cd a) { return {a.y, -a.x}; }   // * (-i)

// Orders LDS traffic between the lanes of ONE wave (lanes run in lockstep; this only stops the compiler
// from moving LDS accesses across the exchange point).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The first operations on loop-invariant per-lane twiddles, as volatile asm: the compiler neither hoists them (it would
// park eight more doubles in registers the frame loop does not have) nor copies their inputs (an empty asm with "+v"
// operands kept the arithmetic in the loop at the price of four 64-bit moves per frame).
__device__ __forceinline__ double vol_add(double a, double b) {
    double r;
    asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vol_sub(double a, double b) {
    double r;
    asm volatile("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vol_mul(double a, double b) {
    double r;
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

// forward radix-4 butterfly, outputs in natural order
__device__ __forceinline__ void radix4(cd& x0, cd& x1, cd& x2, cd& x3) {
    const cd t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = mul_mi(csub(x1, x3));
    x0 = cadd(t0, t2);
    x1 = cadd(t1, t3);
    x2 = csub(t0, t2);
    x3 = csub(t1, t3);
}

// the same butterfly when x1 and x3 are known to be zero (the zero tail of a short hop, see ZT below): two complex
// additions instead of eight
__device__ __forceinline__ void radix4_z13(cd& x0, cd& x1, cd& x2, cd& x3) {
    const cd t0 = cadd(x0, x2), t1 = csub(x0, x2);
    x0 = t0;
    x1 = t1;
    x2 = t0;
    x3 = t1;
}

// Second-stage butterflies of dft16 with the W16^(b c) twiddles folded into their additions (fused multiply-adds with
// constant factors): the products by sqrt(1/2) ride in the sums that consume them, and a pair of full twiddles
// cos (1 - i tan) / sin (1 - i cot) leaves its common cosine (sine) to the four output sums.  y_b = T[b][c] in, X[c + 4 d] out
// in the same registers.  22 + 20 + 22 float64 instructions against 28 + 24 + 28 for "twiddle, then radix-4".
constexpr double kC1 = 0.92387953251128673848;    // cos(pi/8)
constexpr double kS1 = 0.38268343236508978178;    // sin(pi/8)
constexpr double kH = 0.70710678118654752440;     // sqrt(1/2)
constexpr double kT8 = 0.41421356237309504880;    // tan(pi/8)
constexpr double kCT8 = 2.41421356237309504880;   // cot(pi/8)

// c = 1: x1 = W^1 y1 = C1 (1 - i T8) y1,  x2 = W^2 y2 = H (1 - i) y2,  x3 = W^3 y3 = S1 (1 - i CT8) y3
__device__ __forceinline__ void radix4_tw1(cd& y0, cd& y1, cd& y2, cd& y3) {
    const cd u2 = {y2.x + y2.y, y2.y - y2.x};
    const cd t0 = {fma(kH, u2.x, y0.x), fma(kH, u2.y, y0.y)};
    const cd t1 = {fma(-kH, u2.x, y0.x), fma(-kH, u2.y, y0.y)};
    const cd p1 = {fma(kT8, y1.y, y1.x), fma(-kT8, y1.x, y1.y)};
    const cd p3 = {fma(kCT8, y3.y, y3.x), fma(-kCT8, y3.x, y3.y)};
    const cd s = {fma(kT8, p3.x, p1.x), fma(kT8, p3.y, p1.y)};      // (x1 + x3) / C1
    const cd d = {fma(-kT8, p3.x, p1.x), fma(-kT8, p3.y, p1.y)};    // (x1 - x3) / C1
    y0 = cd{fma(kC1, s.x, t0.x), fma(kC1, s.y, t0.y)};
    y1 = cd{fma(kC1, d.y, t1.x), fma(-kC1, d.x, t1.y)};
    y2 = cd{fma(-kC1, s.x, t0.x), fma(-kC1, s.y, t0.y)};
    y3 = cd{fma(-kC1, d.y, t1.x), fma(kC1, d.x, t1.y)};
}

// c = 2: x1 = W^2 y1 = H (1 - i) y1,  x2 = W^4 y2 = -i y2,  x3 = W^6 y3 = H (-1 - i) y3
__device__ __forceinline__ void radix4_tw2(cd& y0, cd& y1, cd& y2, cd& y3) {
    const cd t0 = {y0.x + y2.y, y0.y - y2.x};
    const cd t1 = {y0.x - y2.y, y0.y + y2.x};
    const cd u1 = {y1.x + y1.y, y1.y - y1.x};
    const double m3 = y3.y - y3.x, q3 = y3.x + y3.y;                 // x3 = H (m3, -q3)
    const cd s = {u1.x + m3, u1.y - q3};                             // (x1 + x3) / H
    const cd d = {u1.x - m3, u1.y + q3};                             // (x1 - x3) / H
    y0 = cd{fma(kH, s.x, t0.x), fma(kH, s.y, t0.y)};
    y1 = cd{fma(kH, d.y, t1.x), fma(-kH, d.x, t1.y)};
    y2 = cd{fma(-kH, s.x, t0.x), fma(-kH, s.y, t0.y)};
    y3 = cd{fma(-kH, d.y, t1.x), fma(kH, d.x, t1.y)};
}

// c = 3: x1 = W^3 y1 = S1 (1 - i CT8) y1,  x2 = W^6 y2 = H (-1 - i) y2,  x3 = W^9 y3 = -C1 (1 - i T8) y3
__device__ __forceinline__ void radix4_tw3(cd& y0, cd& y1, cd& y2, cd& y3) {
    const double m2 = y2.y - y2.x, q2 = y2.x + y2.y;                 // x2 = H (m2, -q2)
    const cd t0 = {fma(kH, m2, y0.x), fma(-kH, q2, y0.y)};
    const cd t1 = {fma(-kH, m2, y0.x), fma(kH, q2, y0.y)};
    const cd p1 = {fma(kCT8, y1.y, y1.x), fma(-kCT8, y1.x, y1.y)};
    const cd p3 = {fma(kT8, y3.y, y3.x), fma(-kT8, y3.x, y3.y)};
    const cd s = {fma(-kCT8, p3.x, p1.x), fma(-kCT8, p3.y, p1.y)};   // (x1 + x3) / S1
    const cd d = {fma(kCT8, p3.x, p1.x), fma(kCT8, p3.y, p1.y)};     // (x1 - x3) / S1
    y0 = cd{fma(kS1, s.x, t0.x), fma(kS1, s.y, t0.y)};
    y1 = cd{fma(kS1, d.y, t1.x), fma(-kS1, d.x, t1.y)};
    y2 = cd{fma(-kS1, s.x, t0.x), fma(-kS1, s.y, t0.y)};
    y3 = cd{fma(-kS1, d.y, t1.x), fma(kS1, d.x, t1.y)};
}

// In-register 16-point forward DFT.  Input v[j], j = 4a + b.  Output X[k], k = c + 4d, is left in v[4c + d].
// ZT: inputs v[7] and v[15] are zero (pass 1 of a frame whose blocks end in at least 128 zeros).
template <bool ZT = false>
__device__ __forceinline__ void dft16(cd (&v)[16]) {
#pragma unroll
    for (int b = 0; b < (ZT ? 3 : 4); ++b) radix4(v[b], v[4 + b], v[8 + b], v[12 + b]);
    if (ZT) radix4_z13(v[3], v[7], v[11], v[15]);
    // now v[4c + b] = T[b][c]; the radix-4 over b of row c takes its inputs times W16^(b c)
    radix4(v[0], v[1], v[2], v[3]);
    radix4_tw1(v[4], v[5], v[6], v[7]);
    radix4_tw2(v[8], v[9], v[10], v[11]);
    radix4_tw3(v[12], v[13], v[14], v[15]);
}

// DPP move for reductions whose total is read from lane 63 only: lanes without a source lane keep an unspecified value.
// Lane 63 never is one (quad / half-row / row steps have a source in every lane; row_bcast15 feeds rows 1-3 from lane 15 of
// the row before, row_bcast31 feeds rows 2-3 from lane 31), and what the other rows hold after the two broadcast steps never
// reaches it: its row-2 and row-1 inputs are read in the same instruction that would overwrite them.  With all rows
// enabled and bound_ctrl each step folds into one v_add_f32_dpp / v_max_f32_dpp (masked: a zero, a v_mov_b32_dpp and the add).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes in the order of an xor butterfly (pairs, quads, 8, 16, 32, 64: numpy's pairwise tree for
// the mean|x| partials), as DPP row operations: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror,
// row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3; lane 63 holds the total.  Wave-uniform result.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp0<0xB1, 0xF>(v);
    v += dpp0<0x4E, 0xF>(v);
    v += dpp0<0x141, 0xF>(v);
    v += dpp0<0x140, 0xF>(v);
    v += dpp0<0x142, 0xA>(v);
    v += dpp0<0x143, 0xC>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Maximum over the 64 lanes of values >= 0.  Wave-uniform result.
__device__ __forceinline__ float wave_max_pos(float v) {
    v = fmaxf(v, dpp0<0xB1, 0xF>(v));
    v = fmaxf(v, dpp0<0x4E, 0xF>(v));
    v = fmaxf(v, dpp0<0x141, 0xF>(v));
    v = fmaxf(v, dpp0<0x140, 0xF>(v));
    v = fmaxf(v, dpp0<0x142, 0xA>(v));
    v = fmaxf(v, dpp0<0x143, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Raw buffer resources (V#) for the track reads and the per-frame stores.  gfx950 is a GFX9-family part: word 3 =
// DATA_FORMAT 32 (the value composable_kernel uses for gfx90a / gfx94x / gfx950).  A buffer access beyond num_records
// reads 0 / is dropped, checked per dword, which is exactly the reference's zero padding -- see issue_block.
constexpr int kRsrcWord3 = 0x00020000;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Two float32 values in a register pair: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on both at once.  The file is
// built without the SLP vectoriser (Makefile): left to itself it packs scalar float32 arithmetic whose operands sit in
// unrelated registers and pays two or three v_mov_b32 per packed instruction to line them up (measured: 97 moves per
// frame, 1118 -> 1103 VALU instructions with it off).  What is packed here is packed by type, with the pairs laid out
// the same way from the load to the last use: a block is v2f[8], element jj = samples (2 lane + 128 jj, + 1).
typedef float v2f __attribute__((ext_vector_type(2)));

// One 1024-sample analysis block of frame `fi` (exporter.py:230-237) in the lane layout of the FFT:
// blk[jj] = (samples[2*lane + 128*jj], samples[2*lane + 128*jj + 1]).
// Two phases, so that the loads can fly while another frame is being transformed: issue_block requests the raw
// interleaved samples, mix_block (mono mean) is the first thing that waits for them.
template <int C>
struct RawBlock {
    float q[8][2 * C];   // jj -> two consecutive samples (C channels each)
};

// The frame's read window [pos, pos + nvalid), nvalid = min(hop, 1024, n_samp - pos) (a short read at the end of the
// file, exporter.py:231-236), becomes the buffer resource itself: base = &audio[pos], num_records = nvalid samples.
// Every sample of the block beyond the window -- the 224 zeros of an 800-sample hop, the tail of the last frames, the
// whole block in front of the track start (fi < 0: Analyzer.buffer is zero, analysis.py:14) -- is then out of range and
// loads as 0 without a compare, a select or a clamped address in the vector ALU: the resource is four scalar registers
// built by the scalar ALU from the wave-uniform pos[fi].  Eight unconditional loads of two samples each.
// DEPENDENCY: the per-jj step (128 jj samples) travels in the instruction's soffset operand, and the zero padding relies
// on the range check covering voffset + soffset.  gfx950 (like the rest of the GFX9 family, raw buffers, no swizzle)
// checks `offset = voffset + inst_offset + soffset` against num_records; LLVM's builtin documentation calls soffset
// "excluded from bounds checking" for the sake of other generations.  tests/test_gpu_parity.py::
// test_zero_padding_is_the_buffer_range_check pins the behaviour directly (a track of non-zero samples, with non-zero
// memory behind its end): a toolchain or part that drops the check fails there, loudly.
// ZT (zero tail): the caller knows hop <= 896, so samples 896..1023 of every block (jj = 7 in every lane) are padding:
// they are neither loaded nor mixed nor windowed, and pass 1 of the FFT drops them from its first butterflies.
template <int C, bool ZT>
__device__ __forceinline__ void issue_block(const float* __restrict__ audio, long n_samp,
                                            const long* __restrict__ pos, long fi, int hop, int lane,
                                            RawBlock<C>& raw) {
    const long p = pos[fi < 0 ? 0 : fi];
    const unsigned long rem = (unsigned long)(n_samp - p);          // >= 1: pos is clamped to n_samp - 1
    unsigned nv = (unsigned)(rem >> 32) != 0u ? (unsigned)AUR_BLOCK : (unsigned)rem;
    nv = nv < (unsigned)AUR_BLOCK ? nv : (unsigned)AUR_BLOCK;
    nv = nv < (unsigned)hop ? nv : (unsigned)hop;
    if (fi < 0) nv = 0u;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(audio + p * C), 0, (int)(nv * (unsigned)(C * 4)), kRsrcWord3);
    const int voff = lane * (2 * C * 4);
#pragma unroll
    for (int jj = 0; jj < (ZT ? 7 : 8); ++jj) {
        if (C == 2) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 128 * jj * C * 4, 0);   // (L0, R0, L1, R1)
            raw.q[jj][0] = __uint_as_float(q.x);
            raw.q[jj][1] = __uint_as_float(q.y);
            raw.q[jj][2 * C - 2] = __uint_as_float(q.z);
            raw.q[jj][2 * C - 1] = __uint_as_float(q.w);
        } else {
            const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 128 * jj * C * 4, 0);
            raw.q[jj][0] = __uint_as_float(q.x);
            raw.q[jj][2 * C - 1] = __uint_as_float(q.y);
        }
    }
}

template <int C, bool ZT>
__device__ __forceinline__ void mix_block(const RawBlock<C>& raw, v2f (&blk)[8]) {
    if (ZT) blk[7] = v2f{0.0f, 0.0f};
#pragma unroll
    for (int jj = 0; jj < (ZT ? 7 : 8); ++jj) {
        // y.mean(axis=1) in float32: (L + R) rounded, then * 0.5 (exporter.py:234); padding zeros stay zero.
        // Two scalar additions into a register pair, one packed multiplication.
        if (C == 2)
            blk[jj] = v2f{raw.q[jj][0] + raw.q[jj][1], raw.q[jj][2 * C - 2] + raw.q[jj][2 * C - 1]} * 0.5f;
        else
            blk[jj] = v2f{raw.q[jj][0], raw.q[jj][2 * C - 1]};
    }
}

template <int C, bool ZT>
__device__ __forceinline__ void load_block(const float* __restrict__ audio, long n_samp,
                                           const long* __restrict__ pos, long fi, int hop, int lane,
                                           v2f (&blk)[8]) {
    RawBlock<C> raw;
    issue_block<C, ZT>(audio, n_samp, pos, fi, hop, lane, raw);
    mix_block<C, ZT>(raw, blk);
}

struct Tables {
    const v2f* win;      // LDS: (w[2n], w[2n+1]), n < 1024
    const cd* tw2;       // LDS: W64^(r q), [r][q]
    const cd* tw1;       // LDS: &W1024^(lane k1)[k1 - 1][lane]; rows are 64 complex doubles (ds_read_b128, lane-linear)
    cd wuh;              // 0.5 * W2048^lane
    cd wus;              // the same, except lane 0: 0.5 * W2048^128 (its pairs (640,384), (896,128); see unpack_mag)
    int b0, b1;          // exchange-2 positions of the partner groups: Z[b + 256 s], s < 4
    int off2;            // first bin of the slots (ql = 0, s >= 2): lane, except lane 0: 128
    const double2* lg;   // LDS: (1 / c_i, log c_i), c_i = 1 + i / 128, i <= 128 (log1p_cr)
};

// Stages the window and the W64 table into LDS (whole workgroup), loads the per-lane twiddle bases.  Ends with
// a barrier.
__device__ __forceinline__ Tables stage_tables(const float* __restrict__ window, const double* __restrict__ twid,
                                               unsigned char* smem, int lane) {
    float4* w4 = reinterpret_cast<float4*>(smem);
    const float4* wsrc = reinterpret_cast<const float4*>(window);
    for (int i = threadIdx.x; i < AUR_FFT_SIZE / 4; i += blockDim.x) w4[i] = wsrc[i];
    double* t2 = reinterpret_cast<double*>(smem + kLdsWinBytes);
    for (int i = threadIdx.x; i < 128; i += blockDim.x) t2[2 * kTw2Stride * (i >> 5) + (i & 31)] = twid[AUR_TW_W2 + i];
    Tables t;
    t.win = reinterpret_cast<const v2f*>(smem);
    t.tw2 = reinterpret_cast<const cd*>(t2);
    double2* t1 = reinterpret_cast<double2*>(smem + kLdsWinBytes + kLdsTw2Bytes);
    const double2* t1src = reinterpret_cast<const double2*>(twid + AUR_TW_P1);
    for (int i = threadIdx.x; i < 15 * 64; i += blockDim.x) t1[i] = t1src[i];
    t.tw1 = reinterpret_cast<const cd*>(t1) + lane;
    double2* lg = reinterpret_cast<double2*>(smem + kLdsWinBytes + kLdsTw2Bytes + kLdsTw1Bytes);
    const double2* lgsrc = reinterpret_cast<const double2*>(twid + AUR_TW_LG);
    for (int i = threadIdx.x; i < AUR_LG_ENTRIES; i += blockDim.x) lg[i] = lgsrc[i];
    t.lg = lg;
    t.wuh = cd{twid[AUR_TW_WU + 2 * lane], twid[AUR_TW_WU + 2 * lane + 1]};
    t.wus = cd{twid[AUR_TW_WS + 2 * lane], twid[AUR_TW_WS + 2 * lane + 1]};
    t.b0 = (lane == 0) ? 128 : 256 - lane;
    t.b1 = 192 - lane;
    t.off2 = (lane == 0) ? 128 : lane;
    __syncthreads();
    return t;
}

// mean(|samples|) of one block in numpy's pairwise order (8 blocks of 128, 8 strided accumulators each, then
// the fixed add tree); `xf` is the wave's exchange buffer.  Returns the mean in every lane.
__device__ __forceinline__ float block_meanabs(const v2f (&cur)[8], int lane, float* xf) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
        *reinterpret_cast<v2f*>(xf + 2 * lane + 136 * jj) = cur[jj];   // element i lives at i + 8 * (i / 128)
    wave_sync();
    // |x| is taken where the value is added: a source modifier of v_add_f32 instead of 16 v_and_b32 in front of the stores
    const float* e = xf + 136 * (lane >> 3) + (lane & 7);
    float r = fabsf(e[0]);
#pragma unroll
    for (int t = 1; t < 16; ++t) r += fabsf(e[8 * t]);
    r = wave_sum(r);
    wave_sync();
    return r * (1.0f / 1024.0f);
}

// A block serves two frames: as the second half of the buffer in its own frame and as the first half in the next one
// (analysis.py:19-20).  What is carried from frame to frame is the block times the FIRST half of the window, computed while
// the block is still at hand: the packed products land in the carried registers themselves, where the hand-over
// "previous <- current" used to be seven register moves on top of the same seven products.
template <bool ZT>
__device__ __forceinline__ void window_first_half(const v2f (&blk)[8], int lane, const Tables& T, const v2f* wreg,
                                                  v2f (&pw)[8]) {
#pragma unroll
    for (int j = 0; j < (ZT ? 7 : 8); ++j) pw[j] = blk[j] * (wreg ? wreg[j] : T.win[lane + 64 * j]);
}

// 1024-point complex FFT (float64) of the windowed, packed 2048-sample buffer [previous block | cur].
// In: cur[jj] = block samples (2*lane + 128*jj, +1); pw[jj] = the same samples of the previous block, already times the
// first half of the window (float32, as buffer * window rounds them; see window_first_half).
// Out (slot p = 4 ql + s, ql < 2, s < 4):  v[p] = Z[lane + 64 (ql + 4 s)],  v[8 + p] = Z[b_ql + 256 s] with
// b_0 = 256 - lane, b_1 = 192 - lane, i.e. v[8 + 4 ql + (3 - s)] = Z[1024 - (lane + 64 (ql + 4 s))], the unpack
// partner of v[p].  Lane 0 (whose bins 64 j pair among themselves) reads b_0 = 128 instead: v[8 + s] = Z[128 + 256 s].
template <bool ZT = false>
__device__ __forceinline__ void fft_packed(const v2f (&pw)[8], const v2f (&cur)[8], int lane,
                                           const Tables& T, double* xb_, cd (&v)[16], const v2f* wreg = nullptr,
                                           const cd* tw3reg = nullptr, const cd* tw1reg = nullptr) {
    // volatile: keeps one ds_read_b64 / ds_write_b64 per element.  Merged into ds_read2_b64 / ds_write2_b64 the
    // accesses fall back to 32-bank addressing (half the LDS rate) and the padded rows conflict two ways.
    typedef __attribute__((address_space(3))) volatile double lds_vdouble;
    lds_vdouble* xb = (lds_vdouble*)xb_;
    // window in float32 exactly as the reference (buffer * window, analysis.py:25), then widen and pack:
    // z[l + 64 j] = (x[2n] w[2n], x[2n+1] w[2n+1])
#pragma unroll
    for (int j = 0; j < (ZT ? 7 : 8); ++j) v[j] = cd{(double)pw[j].x, (double)pw[j].y};
#pragma unroll
    for (int j = 0; j < (ZT ? 7 : 8); ++j) {
        const v2f xw = cur[j] * (wreg ? wreg[j + 8] : T.win[lane + 64 * (j + 8)]);
        v[j + 8] = cd{(double)xw.x, (double)xw.y};
    }
    if (ZT) v[7] = v[15] = cd{0.0, 0.0};
    // pass 1: radix-16 over j, twiddle W1024^(l k1), transpose through LDS (rows padded to 66: conflict-free)
    dft16<ZT>(v);
    // twiddle W1024^(l k1) from the LDS table (one ds_read_b128 per value: the LDS array has cycles to spare, the
    // float64 VALU does not -- building the powers per frame cost 56 of its instructions)
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) {
        const int i = 4 * (k1 & 3) + (k1 >> 2);     // register that holds output k1 of dft16
        v[i] = cmul(v[i], (tw1reg && k1 <= AUR_STFT_TW1_REGS) ? tw1reg[k1] : T.tw1[64 * (k1 - 1)]);
    }
    const int lo4 = lane & 15, hi2 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) xb[lane + 66 * ((i >> 2) + 4 * (i & 3))] = v[i].x;
    wave_sync();
#pragma unroll
    for (int p = 0; p < 16; ++p) v[p].x = xb[(hi2 + 4 * p) + 66 * lo4];
    wave_sync();
#pragma unroll
    for (int i = 0; i < 16; ++i) xb[lane + 66 * ((i >> 2) + 4 * (i & 3))] = v[i].y;
    wave_sync();
#pragma unroll
    for (int p = 0; p < 16; ++p) v[p].y = xb[(hi2 + 4 * p) + 66 * lo4];
    wave_sync();
    // pass 2: radix-16 over p (lane = k1 + 16 r), transpose through LDS; the twiddle W64^(r q) waits behind the
    // transpose, where r is the register index and r = 0 needs none: 12 products per lane instead of 15
    dft16(v);
#pragma unroll
    for (int i = 0; i < 16; ++i) xb[lo4 + 16 * ((i >> 2) + 4 * (i & 3)) + 256 * hi2] = v[i].x;
    wave_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i].x = xb[lane + 64 * (i >> 2) + 256 * (i & 3)];   // i = 4 ql + r
#pragma unroll
    for (int i = 0; i < 4; ++i) v[8 + i].x = xb[T.b0 + 256 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[12 + i].x = xb[T.b1 + 256 * i];
    wave_sync();
#pragma unroll
    for (int i = 0; i < 16; ++i) xb[lo4 + 16 * ((i >> 2) + 4 * (i & 3)) + 256 * hi2] = v[i].y;
    wave_sync();
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i].y = xb[lane + 64 * (i >> 2) + 256 * (i & 3)];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[8 + i].y = xb[T.b0 + 256 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[12 + i].y = xb[T.b1 + 256 * i];
    wave_sync();
    // pass 3: radix-4 over r.  Position k1 + 16 q + 256 r of the exchange holds pass-2 output q of lane k1 + 16 r;
    // the radix-4 over r of the group at base k1 + 16 q, inputs times W64^(r q), yields Z[base + 256 s], s < 4.
    // Group g of a lane sits at base lane, lane + 64, b0, b1: q = (base >> 4) & 15 is a per-lane constant, so the
    // twelve twiddles are loop invariants (tw3reg: registers of the frame kernel; else read from the LDS table).
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int q = ((g == 0 ? lane : g == 1 ? lane + 64 : g == 2 ? T.b0 : T.b1) >> 4) & 15;
#pragma unroll
        for (int r = 1; r < 4; ++r)
            v[4 * g + r] = cmul(v[4 * g + r], tw3reg ? tw3reg[3 * g + r - 1] : T.tw2[r * kTw2Stride + q]);
        radix4(v[4 * g + 0], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    }
}

// |X| in float32 from |X|^2 in float64: |X|^2 rounded to float32 (relative 2^-24; never outside the float32 range
// for audio in [-1, 1]: |X| <= 2048, and leakage bins stop at the ~1e-7 rounding noise of the float32 window
// product) and v_sqrt_f32 (1 ulp): within 1.5 ulp of the exact magnitude at a quarter of v_sqrt_f64's cost.
__device__ __forceinline__ float mag_from_sq(double m2) { return __builtin_amdgcn_sqrtf((float)m2); }

// log1p for x >= 0 in float32 (the contract is rtol 1e-4; np.log1p is within 1 ulp; this is within ~4 ulp):
// ln2 * log2(u) + c / u with u = fl(1 + x) and c = x - (u - 1) the rounding error of u.  c / u only matters next to
// u = 1 (it is below 2^-24 of the result once u >= 2), where 1 / u = 2 - u up to (u - 1)^2: c * max(2 - u, 0) replaces
// the reciprocal.  v_log_f32 is relatively accurate next to 1 (measured over the whole range by
// tests/test_gpu_parity.py::test_log1p_probe), and for u == 1 the result is c = x exactly.
__device__ __forceinline__ float log1p_pos(float x) {
    const float u = 1.0f + x;
    const float c = x - (u - 1.0f);
    return fmaf(c, fminf(fmaxf(2.0f - u, 0.0f), 1.0f), __builtin_amdgcn_logf(u) * 0.693147180559945309f);   // clamp modifier
}

// ---- the displayed bins (k < 128; the export only draws spectrum[:80], rt_widget.py:935) in the reference's own chain ----
//
// A bar top is int(v * (h - 10) * ...) (rt_widget.py:986) and a particle row int(h / 2 + (v - 0.5) * h * 0.6) (:997), so a
// last-ulp difference in log1p|X| moves them by a row in a fraction of the frames.  For bins < 128 the magnitude is therefore
// computed the way numpy computes np.abs of the complex64 spectrum (audio/analysis.py:26) on the x86 hosts the fixtures
// come from (numpy 2.x, AVX2 / AVX-512 + FMA3 dispatch of loops_unary_complex): round re, im to float32, then in float32
//   |z| = max * sqrt(fma(d, d, 1)),  d = min / max,  max = max(|re|, |im|), min = min(|re|, |im|)
// with IEEE division and square root -- bit-identical to np.abs(complex64) on 2 000 000 random values
// (tests/test_host_cpu.py::test_numpy_cabs_formula).  The other bins keep mag_from_sq (flux only needs rtol 1e-4).
// sqrt of x in [1, 2], correctly rounded: v_sqrt_f32 (1 ulp) and the residual test of its two neighbours -- the steps of the
// compiler's own IEEE expansion of sqrtf without the input scaling and the 0 / infinity handling this range never needs
// (7 of its 16 instructions).
__device__ __forceinline__ float sqrt_cr_1to2(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __int_as_float(__float_as_int(s) - 1), up = __int_as_float(__float_as_int(s) + 1);
    const float rd = fmaf(-dn, s, x), ru = fmaf(-up, s, x);
    const float r = rd <= 0.0f ? dn : s;
    return ru > 0.0f ? up : r;
}

__device__ __forceinline__ float cabs_c64(double re_d, double im_d) {
    const float a = fabsf((float)re_d), b = fabsf((float)im_d);      // astype(complex64)
    const float mx = fmaxf(a, b), mn = fminf(a, b);
    const float d = mx > 0.0f ? mn / mx : 0.0f;                      // correctly rounded (no fast-math on this library)
    return mx * sqrt_cr_1to2(fmaf(d, d, 1.0f));                      // d in [0, 1]
}

// Correctly rounded log1p of a float32 x >= 0 (np.log1p at audio/analysis.py:33 is within 1-2 ulp of this: numpy's
// AVX-512 build calls SVML, whose rounding cannot be reproduced; 92.3 % of its results are the correctly rounded value).
// float64: u = 1 + x is exact for x >= 2^-29; u = 2^e m, m in [1, 2); c = nearest multiple of 1/128 to m;
// log u = e ln2 + log c + log1p(r), r = m / c - 1, |r| <= 2^-8 (degree-5 series: 2^-48 / 6 absolute); c = 1 gives
// log c = 0 exactly, so results next to 0 keep their relative accuracy.  Below 2^-20: x - x^2 / 2.
__device__ __forceinline__ float log1p_cr(float x, const double2* __restrict__ lg) {
    const double xd = (double)x;
    const double u = 1.0 + xd;
    const unsigned long long ub = (unsigned long long)__double_as_longlong(u);
    const unsigned hi = (unsigned)(ub >> 32);
    const int e = (int)(hi >> 20) - 1023;
    const unsigned i = (((hi >> 12) & 0xffu) + 1u) >> 1;             // 0 .. 128
    const double m = __longlong_as_double((long long)((ub & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    const double2 t = lg[i];
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, 0.2, -0.25);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    p = fma(r * r, p, r);
    const double big = fma((double)e, 0.693147180559945309417, t.y) + p;
    const double small = fma(-0.5 * xd, xd, xd);
    return (float)(x < 9.5367431640625e-7f ? small : big);
}

// W32^j = exp(-2 pi i j / 32) for j = 1, 5
constexpr double kW32re1 = 0.98078528040323044913, kW32im1 = -0.19509032201612826785;
constexpr double kW32re5 = 0.55557023301960222474, kW32im5 = -0.83146961230254523708;

// Per-lane spectrum state.  Slot p = 4 ql + s (ql < 2, s < 4) of lane l is the bin pair (kA, 1024 - kA) with
// kA = l + 64 (ql + 4 s); in lane 0 the slots (ql = 0, s >= 2) are kA = 640, 896 instead (see unpack_mag).  The 512
// pairs tile bins 0..1024 except bin 512, which lane 0 owns as "mid".
__device__ __forceinline__ int slot_bin(int lane, int p) {
    const int ql = p >> 2, s = p & 3;
    return lane + 64 * (ql + 4 * s) + ((lane == 0 && ql == 0 && s >= 2) ? 128 : 0);
}


struct Mags {
    v2f ab[8];
    float mid;
    // m - previous, then previous <- m as ONE 64-bit move into the register pair the previous value lived in: the tied
    // operand keeps the pair in place, the unused operand d orders the move behind the subtraction (left to the compiler the
    // hand-over was two v_mov_b32 per slot at the loop head).
    // The s_nop is NOT optional: m comes out of v_sqrt_f32, and on gfx940 / gfx950 a non-transcendental VALU instruction
    // that reads the result of a transcendental one needs one wait state in between.  The compiler's hazard recogniser
    // inserts it for its own instructions but does not look inside an asm statement: without the nop the halo frame of a
    // chunk (where the scheduler happened to put the move right behind the square root) handed stale magnitudes on and the
    // first flux of every chunk was off by a few per cent (caught by test_analysis_vs_reference_fixtures).
    __device__ __forceinline__ v2f diff(int p, v2f m) {
        const v2f d = m - ab[p];
        asm("s_nop 0\n\tv_mov_b64 %0, %1" : "+v"(ab[p]) : "v"(m), "v"(d));
        return d;
    }
    __device__ __forceinline__ float swap_mid(float m) { const float t = mid; mid = m; return t; }
};

// The same state parked in LDS (float [17][64] per wave, lane-private columns: no conflicts, no synchronisation):
// 17 registers fewer across the whole frame for 34 LDS accesses per frame.  Build knob AUR_STFT_PMAG_LDS.
struct MagsLds {
    float* col;          // &area[lane]
    __device__ __forceinline__ v2f diff(int p, v2f m) {
        const v2f t = {col[64 * (2 * p)], col[64 * (2 * p + 1)]};
        col[64 * (2 * p)] = m.x;
        col[64 * (2 * p + 1)] = m.y;
        return m - t;
    }
    __device__ __forceinline__ float swap_mid(float m) { const float t = col[64 * 16]; col[64 * 16] = m; return t; }
};

// log1p_pos on both bins of a slot (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32; the logarithms stay scalar)
__device__ __forceinline__ v2f log1p_pos2(v2f x) {
    const v2f u = 1.0f + x;
    const v2f c = x - (u - 1.0f);
    // u >= 1, so 2 - u <= 1 and max(2 - u, 0) is the clamp of 2 - u to [0, 1]: the output modifier of the packed
    // subtraction (the compiler folds it for the scalar form in log1p_pos, not for v_pk_add_f32: two v_max_f32 per pair)
    v2f t;
    asm("v_pk_add_f32 %0, %1, 2.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0] clamp" : "=v"(t) : "v"(u));
    const v2f lg = {__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)};
    return __builtin_elementwise_fma(c, t, lg * 0.693147180559945309f);
}

// Real-FFT unpack -> |X| -> flux partial vs pm (updated in place) -> log1p.
// X[kA] = E + T and X[kB] = conj(E - T), kB = 1024 - kA, share E = (Z[kA] + conj Z[kB]) / 2 and
// T = W2048^kA * (-i) (Z[kA] - conj Z[kB]) / 2, so a lane computes both bins of a pair from one product.
// fft_packed leaves both operands of a pair in the lane: slot p = 4 ql + s pairs v[p] = Z[kA] with
// v[8 + 4 ql + (3 - s)] = Z[kB].  Lane 0 is the exception for ql = 0: its bins 0, 128, .., 1024 pair among themselves.
// It holds Z[256 s] in v[s] and Z[128 + 256 s] in v[8 + s] and wires its four slots as
//   s = 0: (Z[0], Z[0]) -> X[0], X[1024]      s = 2: (Z[640], Z[384])   (generic kA shifted by 128)
//   s = 1: (Z[256], Z[768])                   s = 3: (Z[896], Z[128])   (generic kA shifted by 128)
// by four complex register moves executed with lane 0 alone enabled (lane0_rewire; as selects in every lane they were 32
// v_cndmask_b32 per frame); X[512] = conj Z[512] = conj v[2], read before the moves, is the extra "mid" bin.
// Twiddles th = 0.5 W2048^kA = wuh W32^j, j = ql + 4 s: W32^4 = (1 - i) / sqrt 2, W32^8 = -i, so only W32^1 and
// W32^5 need a full product; T.wus equals T.wuh except in lane 0, where it carries the shift by 128 bins.
// Returns this lane's flux partial.  When EMIT: lg0 / lg1 receive log1p|X| of bins lane and lane + 64 (band maxima),
// and when STORE_LOG too, log1p|X| is stored to `out` (float32[1025]) as soon as it is known.
// `out` is the frame's float32[1025] spectrum as a buffer resource (base = the frame's row, num_records = 4100 bytes).
// A lane's 17 values are 17 different 256-byte pieces of the row, and 17 buffer_store_dword per frame are what a wave pays
// for in ISSUE, not in bandwidth (round 3: without the stores the kernel ran 15 % faster, with all stores aimed at 1 024
// cache-resident rows it did not: profiles/r03_stft_lesions.txt).  So the row is first put together in the wave's
// exchange buffer `stage` (free between the last exchange and the next frame: 17 ds_write_b32, lane-linear both ways)
// and leaves as four buffer_store_dwordx4 plus one dword -- build knob AUR_STFT_WIDE_STORES=1.  Measured 0.82 ms against
// 0.78 ms for the direct stores (profiles/r03_stft_lesions.txt): the two extra wave-wide LDS round trips per frame cost
// more than the twelve store instructions they save, so the direct stores stay the default.
// Lane 0 only: v[2] <- v[10] (Z[640]), v[10] <- v[3] (Z[768]), v[3] <- v[11] (Z[896]), v[11] <- v[0] (Z[0]), after which
// the generic wiring of the slots (A = v[s], P = v[11 - s]) reads lane 0's own pairs.  One asm block: EXEC is narrowed
// to lane 0 and restored inside it, so the frame body stays one basic block.  The whole wave is active here (the only exit
// above is wave-uniform).
__device__ __forceinline__ void lane0_rewire(cd (&v)[16]) {
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "v_mov_b64 %[v2x], %[v10x]\n\t"
        "v_mov_b64 %[v2y], %[v10y]\n\t"
        "v_mov_b64 %[v10x], %[v3x]\n\t"
        "v_mov_b64 %[v10y], %[v3y]\n\t"
        "v_mov_b64 %[v3x], %[v11x]\n\t"
        "v_mov_b64 %[v3y], %[v11y]\n\t"
        "v_mov_b64 %[v11x], %[v0x]\n\t"
        "v_mov_b64 %[v11y], %[v0y]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sv] "=&s"(saved), [v2x] "+v"(v[2].x), [v2y] "+v"(v[2].y), [v10x] "+v"(v[10].x), [v10y] "+v"(v[10].y),
          [v3x] "+v"(v[3].x), [v3y] "+v"(v[3].y), [v11x] "+v"(v[11].x), [v11y] "+v"(v[11].y)
        : [v0x] "v"(v[0].x), [v0y] "v"(v[0].y));
}

template <bool EMIT, bool STORE_LOG, typename PM, typename MID>
__device__ __forceinline__ float unpack_mag(cd (&v)[16], int lane, const Tables& T, PM& pm,
                                            const __amdgpu_buffer_rsrc_t out, float* stage, float& lg0, float& lg1,
                                            MID&& mid_hook, const cd* threg = nullptr) {
    constexpr double H = 0.70710678118654752440;
    const bool l0 = lane == 0;
    const cd wuh = T.wuh, wus = T.wus;
    cd th[8];
    th[0] = wuh;                                                       // j = 0
    th[1] = cd{H * vol_add(wuh.x, wuh.y), H * vol_sub(wuh.y, wuh.x)};  // j = 4
    th[2] = mul_mi(wus);                                               // j = 8
    th[3] = mul_mi(cd{H * vol_add(wus.x, wus.y), H * vol_sub(wus.y, wus.x)});      // j = 12
    // cmul(wuh, W32^1) and cmul(wuh, W32^5) with their leading products kept in the loop
    if (threg) {
        th[4] = threg[0];
        th[5] = threg[1];
    } else {
        th[4] = cd{fma(wuh.x, kW32re1, -vol_mul(wuh.y, kW32im1)), fma(wuh.x, kW32im1, vol_mul(wuh.y, kW32re1))};   // j = 1
        th[5] = cd{fma(wuh.x, kW32re5, -vol_mul(wuh.y, kW32im5)), fma(wuh.x, kW32im5, vol_mul(wuh.y, kW32re5))};   // j = 5
    }
    th[6] = mul_mi(th[4]);                                             // j = 9
    th[7] = mul_mi(th[5]);                                             // j = 13
    v2f fl2 = {0.0f, 0.0f};
    // bin 512 (lane 0 owns it): X[512] = conj Z[512]; its logarithm rides with slot 0's second bin
    const float mm = mag_from_sq(fma(v[2].x, v[2].x, v[2].y * v[2].y));
    float lmid = 0.0f;
    lane0_rewire(v);
    // byte offsets of the stores: bin lane + 64 j = off_a + 256 j; bin 1024 - lane - 64 j = off_b + (3840 - 256 j) with
    // off_b = 256 - 4 lane > 0; the shifted slots of lane 0 (bins off2 + 64 j and 1024 - off2 - 64 j, j = 8, 12) use
    // off_a2 = 4 off2 and off_b2 = 4 (256 - off2) + 4 (768 - 64 j)
    const int off_a = 4 * lane, off_b = 256 - 4 * lane;
    const int off_a2 = 4 * T.off2, off_b2 = 4 * (256 - T.off2);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int ql = p >> 2, s = p & 3, j = ql + 4 * s;
        if (p == AUR_STFT_MID_AT) mid_hook();   // half of v[] is dead from here on
        const cd A = v[p], P = v[8 + 4 * ql + (3 - s)];
        const double a = A.x, b = A.y, c = P.x, d = P.y;
        const double ex = a + c, ey = b - d;     // 2E
        const double ox = b + d, oy = c - a;     // 2O = -i (Z[kA] - conj Z[kB])
        const double tx = fma(th[p].x, ox, -(th[p].y * oy));
        const double ty = fma(th[p].x, oy, th[p].y * ox);
        const double ar = fma(0.5, ex, tx), ai = fma(0.5, ey, ty);
        const double br = fma(0.5, ex, -tx), bi = fma(0.5, ey, -ty);
        // j < 2: bin kA = lane + 64 j is one of the displayed bins: the reference's own magnitude / log1p chain
        const v2f m = {j < 2 ? cabs_c64(ar, ai) : mag_from_sq(fma(ar, ar, ai * ai)), mag_from_sq(fma(br, br, bi * bi))};
        fl2 += __builtin_elementwise_max(pm.diff(p, m), (v2f)(0.0f));
        if (EMIT) {
            v2f l;
            if (p == 0) {
                const v2f lp = log1p_pos2(v2f{m.y, mm});
                l = v2f{log1p_cr(m.x, T.lg), lp.x};
                lmid = lp.y;
            } else {
                l = j < 2 ? v2f{log1p_cr(m.x, T.lg), log1p_pos(m.y)} : log1p_pos2(m);
            }
            if (j == 0) lg0 = l.x;
            if (j == 1) lg1 = l.x;
            if (STORE_LOG) {
                // byte offsets inside the row (= inside the staged copy of it)
                const int oa = (p == 2 || p == 3) ? off_a2 + 256 * j : off_a + 256 * j;                    // bin kA
                const int ob = (p == 2 || p == 3) ? off_b2 + 4 * (768 - 64 * j) : off_b + (3840 - 256 * j);   // bin 1024 - kA
                if (AUR_STFT_WIDE_STORES) {
                    stage[oa >> 2] = l.x;
                    stage[ob >> 2] = l.y;
                } else {
                    aur_buf_st(l.x, out, oa);
                    aur_buf_st(l.y, out, ob);
                }
            }
        }
    }
    float fl = fl2.x + fl2.y;
    if (l0) {   // bin 512
        fl += fmaxf(0.0f, mm - pm.swap_mid(mm));
        if (EMIT && STORE_LOG) {
            if (AUR_STFT_WIDE_STORES) stage[512] = lmid;
            else aur_buf_st(lmid, out, 4 * 512);
        }
    }
    if (EMIT && STORE_LOG && AUR_STFT_WIDE_STORES) {
        wave_sync();                                           // the row is complete in LDS
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(stage + 4 * lane + 256 * i);
            __builtin_amdgcn_raw_buffer_store_b128(q, out, 16 * lane + 1024 * i, 0, AUR_STFT_NT ? 2 : 0);
        }
        if (l0) aur_buf_st(stage[1024], out, 4096);
        wave_sync();                                           // nothing of the next frame touches the buffer before these reads
    }
    return fl;
}

__device__ __forceinline__ void mags_zero(Mags& m) {
#pragma unroll
    for (int p = 0; p < 8; ++p) m.ab[p] = v2f{0.0f, 0.0f};
    m.mid = 0.0f;
}

__device__ __forceinline__ void mags_load(Mags& m, const float* __restrict__ src, int lane) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int k = slot_bin(lane, p);
        m.ab[p] = v2f{src[k], src[1024 - k]};
    }
    m.mid = src[512];
}

__device__ __forceinline__ void mags_store(const Mags& m, float* __restrict__ dst, int lane) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int k = slot_bin(lane, p);
        dst[k] = m.ab[p].x;
        dst[1024 - k] = m.ab[p].y;
    }
    if (lane == 0) dst[512] = m.mid;
}

// LOG: logspec is stored; BM: band maxima are stored; SAMPLES: the mono blocks are stored (waveform mode).
// ZT: hop <= 896, every block ends in at least 128 zeros (48 kHz at 60 frames/s: hop 800).
template <int C, bool LOG, bool BM, bool SAMPLES, bool ZT>
__global__ __launch_bounds__(kWavesPerWG * 64, AUR_STFT_WAVES_PER_SIMD) void stft_frames_kernel(
    const float* __restrict__ audio, long n_samp, const long* __restrict__ pos, long frame_begin, int n_frames,
    int hop, int frames_per_wave, const float* __restrict__ window, const double* __restrict__ twid,
    float* __restrict__ logspec, float* __restrict__ flux_sum, float* __restrict__ meanabs,
    float* __restrict__ samples_out, float* __restrict__ bandmax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    // readfirstlane: tells the compiler the wave index is wave-uniform, so the frame range, every per-frame pointer
    // and pos[f] live in scalar registers and the per-frame branches are scalar branches
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Tables T = stage_tables(window, twid, smem, lane);
    // The per-lane twiddles come from global memory and are first used inside the frame loop.  A use here makes the
    // compiler wait for those loads in front of the loop; otherwise its wait-count pass, which cannot tell the first
    // iteration from the others, puts a full s_waitcnt vmcnt(0) in front of their first use INSIDE the loop, and every frame
    // then waits for the previous frame's spectrum stores to be acknowledged.
    asm volatile("" ::"v"(T.wuh.x), "v"(T.wuh.y), "v"(T.wus.x), "v"(T.wus.y));

    double* xb = reinterpret_cast<double*>(smem + kLdsTabBytes) + wave * kXbufDoubles;   // wave-private
    float* xf = reinterpret_cast<float*>(xb);

    const long gw = (long)blockIdx.x * kWavesPerWG + wave;
    const long frame_end = frame_begin + n_frames;
    const long c0 = frame_begin + gw * frames_per_wave;
    if (c0 >= frame_end) return;   // no block-level barrier below this point
    const long cend = (c0 + frames_per_wave < frame_end) ? c0 + frames_per_wave : frame_end;

#if AUR_STFT_PMAG_LDS
    MagsLds pmag;                         // Analyzer.prev_mag of this wave's bins, lane-private LDS columns
    pmag.col = reinterpret_cast<float*>(smem + kLdsTabBytes + kWavesPerWG * kXbufDoubles * 8) + wave * 17 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 17; ++i) pmag.col[64 * i] = 0.0f;
#else
    Mags pmag;                            // Analyzer.prev_mag of this wave's bins
    mags_zero(pmag);
#endif
    cd v[16];
    cd tw1reg[16];
    if (AUR_STFT_TW1_REGS) {
#pragma unroll
        for (int k1 = 1; k1 <= AUR_STFT_TW1_REGS; ++k1) tw1reg[k1] = T.tw1[64 * (k1 - 1)];
    }
    cd tw3reg[12];                        // W64^(r q_g), g < 4, r = 1..3 (fft_packed, pass 3)
    if (AUR_STFT_TW3_REGS) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = ((g == 0 ? lane : g == 1 ? lane + 64 : g == 2 ? T.b0 : T.b1) >> 4) & 15;
#pragma unroll
            for (int r = 1; r < 4; ++r) tw3reg[3 * g + r - 1] = T.tw2[r * kTw2Stride + q];
        }
    }
    // (only the zero-tail variant has the registers: it carries 7 block pairs instead of 8)
    constexpr bool kThRegs = AUR_STFT_TH_REGS && ZT;
    cd threg[2];                          // the same arithmetic as unpack_mag's own (bit-identical twiddles either way)
    if (kThRegs) {
        threg[0] = cd{fma(T.wuh.x, kW32re1, -(T.wuh.y * kW32im1)), fma(T.wuh.x, kW32im1, T.wuh.y * kW32re1)};
        threg[1] = cd{fma(T.wuh.x, kW32re5, -(T.wuh.y * kW32im5)), fma(T.wuh.x, kW32im5, T.wuh.y * kW32re5)};
    }
    v2f wreg[16];
    if (AUR_STFT_WIN_REGS) {
#pragma unroll
        for (int j = 0; j < 16; ++j) wreg[j] = T.win[lane + 64 * j];
    }

    // Analyzer.buffer starts at zero (analysis.py:14); a chunk that starts inside the track first replays frame
    // c0 - 1 (the halo) to rebuild Analyzer.prev_mag, storing nothing.
    const long fstart = (c0 > 0) ? c0 - 1 : c0;
    v2f pw[8], cur[8];
    load_block<C, ZT>(audio, n_samp, pos, fstart - 1, hop, lane, cur);
    window_first_half<ZT>(cur, lane, T, AUR_STFT_WIN_REGS ? wreg : nullptr, pw);
    load_block<C, ZT>(audio, n_samp, pos, fstart, hop, lane, cur);
    // the same for the first block (mono: the loaded registers ARE the block, first read inside the loop)
#pragma unroll
    for (int jj = 0; jj < (ZT ? 7 : 8); ++jj) asm volatile("" ::"v"(cur[jj]));

    // One frame.  STORE = false is the halo frame: transform and magnitudes only.  The next frame's block is
    // requested after the FFT, before anything of this frame is stored, and is first touched at the end of the body:
    // gfx950 counts loads and stores in one in-order counter (vmcnt), so a load issued behind the 17 spectrum stores
    // of a frame could only be waited for by draining those stores first.
    // (Alternating the roles of previous / current block over a loop unrolled by two instead of handing 16 registers over was
    // measured: 27 VALU instructions fewer per frame, twice the code, 0.841 vs 0.828 ms -- not kept.)
    auto frame = [&](long f, auto store_tag) {
        constexpr bool STORE = decltype(store_tag)::value;
        const long fo = f - frame_begin;
        if (STORE) {
            const float ma = block_meanabs(cur, lane, xf);
            if (lane == 0) meanabs[fo] = ma;
            if (SAMPLES) {
                const __amdgpu_buffer_rsrc_t srs =
                    __builtin_amdgcn_make_buffer_rsrc((void*)(samples_out + fo * AUR_BLOCK), 0, AUR_BLOCK * 4, kRsrcWord3);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(cur[jj].x), __float_as_uint(cur[jj].y)},
                                                          srs, 8 * lane + 512 * jj, 0, 0);
            }
        }
        fft_packed<ZT>(pw, cur, lane, T, xb, v, AUR_STFT_WIN_REGS ? wreg : nullptr,
                       AUR_STFT_TW3_REGS ? tw3reg : nullptr, AUR_STFT_TW1_REGS ? tw1reg : nullptr);
        RawBlock<C> nxt;
        float lg0 = 0.0f, lg1 = 0.0f;
        // the frame's spectrum row as a buffer resource (an empty one when nothing is stored: logspec may be NULL)
        const bool row = STORE && LOG;
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(logspec + (row ? fo * AUR_NBINS : 0)), 0, row ? AUR_NBINS * 4 : 0, kRsrcWord3);
        float fl = unpack_mag<STORE && (LOG || BM), STORE && LOG>(
            v, lane, T, pmag, ors, xf, lg0, lg1,
            [&]() { issue_block<C, ZT>(audio, n_samp, pos, (f + 1 < cend) ? f + 1 : -1, hop, lane, nxt); },
            kThRegs ? threg : nullptr);
        if (STORE) {
            fl = wave_sum(fl);
            if (lane == 0) flux_sum[fo] = fl;
            if (BM) {
                // bins < 80: j = 0 (all lanes) and j = 1 (lanes < 16); bins < 50: j = 0, lanes < 50.  log1p >= 0.
                const float bm80 = wave_max_pos(fmaxf(lg0, (lane < 16) ? lg1 : 0.0f));
                const float bm50 = wave_max_pos((lane < 50) ? lg0 : 0.0f);
                if (lane == 0) {
                    bandmax[2 * fo] = bm80;
                    bandmax[2 * fo + 1] = bm50;
                }
            }
        }
        window_first_half<ZT>(cur, lane, T, AUR_STFT_WIN_REGS ? wreg : nullptr, pw);
        mix_block<C, ZT>(nxt, cur);
    };

    long f = fstart;
    if (c0 > 0) {
        frame(f, std::false_type{});
        ++f;
    }
    for (; f < cend; ++f) frame(f, std::true_type{});
}

// One Analyzer.compute step (audio/analysis.py:25-33) on an already-slid 2048-sample buffer: one wave.
__global__ __launch_bounds__(64) void stft_buffer_kernel(const float* __restrict__ buffer,
                                                         float* __restrict__ prev_mag,
                                                         const float* __restrict__ window,
                                                         const double* __restrict__ twid,
                                                         float* __restrict__ logspec,
                                                         float* __restrict__ flux_sum) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsTabBytes + kXbufDoubles * 8];
    const int lane = threadIdx.x;
    const Tables T = stage_tables(window, twid, smem, lane);
    double* xb = reinterpret_cast<double*>(smem + kLdsTabBytes);
    v2f pw[8], cur[8];
    Mags pmag;
    cd v[16];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        pw[jj] = *reinterpret_cast<const v2f*>(buffer + 2 * lane + 128 * jj) * T.win[lane + 64 * jj];
        cur[jj] = *reinterpret_cast<const v2f*>(buffer + AUR_BLOCK + 2 * lane + 128 * jj);
    }
    mags_load(pmag, prev_mag, lane);
    fft_packed(pw, cur, lane, T, xb, v);
    float lg0, lg1;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)logspec, 0, AUR_NBINS * 4, kRsrcWord3);
    float fl = unpack_mag<true, true>(v, lane, T, pmag, ors, reinterpret_cast<float*>(xb), lg0, lg1, []() {});
    mags_store(pmag, prev_mag, lane);
    fl = wave_sum(fl);
    if (lane == 0) flux_sum[0] = fl;
}

__global__ __launch_bounds__(256) void log1p_probe_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = log1p_pos(x[i]);
}

// Mono mix of a track with more than two channels: y.mean(axis=1) of a float32 [n, C] array (exporter.py:233-234) in
// numpy's own order -- add.reduce over the contiguous axis starts from 0 and, for C < 8, adds the channels left to right;
// from 8 channels on it keeps eight strided accumulators, joins them as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and adds the
// remaining C % 8 channels one by one (pairwise_sum, one block: C <= 128) -- then divides by C in float32.
__global__ __launch_bounds__(256) void mix_channels_kernel(const float* __restrict__ audio, long n_samp, int C,
                                                           float* __restrict__ mono) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_samp) return;
    const float* a = audio + i * C;
    float res;
    if (C < 8) {
        res = 0.0f;
        for (int c = 0; c < C; ++c) res += a[c];
    } else {
        float r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int c = 8;
        for (; c < C - (C % 8); c += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += a[c + k];
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; c < C; ++c) res += a[c];
        res = 0.0f + res;
    }
    mono[i] = res / (float)C;
}

template <int C, bool LOG, bool BM, bool SAMPLES, bool ZT>
int launch_stft_z(aur_ctx* ctx, const float* d_audio, int64_t n_samples, const int64_t* d_pos, int64_t frame_begin,
                int n_frames, int hop, float* d_logspec, float* d_flux_sum, float* d_meanabs, float* d_samples,
                float* d_bandmax, hipStream_t st) {
    // frames per wave: the range is cut into k rounds of one wave per wave slot (2 per SIMD), k as small as keeps a
    // wave at <= 128 frames: long chunks amortise the 1-frame halo, whole rounds leave no tail (216 000 frames: one
    // round of 106; measured 0.764 ms against 0.772 for two rounds of 53, and 0.94 for 160 -- more than one round's worth
    // per wave leaves slots idle)
    const long slots = (long)ctx->num_cus * AUR_STFT_WAVES_PER_CU;
    const long rounds = (n_frames + slots * 128 - 1) / (slots * 128);
    int fpw = (int)((n_frames + slots * rounds - 1) / (slots * rounds));
    if (fpw < 4) fpw = 4;
    if (ctx->stft_fpw > 0) fpw = ctx->stft_fpw;
    const long n_waves = ((long)n_frames + fpw - 1) / fpw;
    const unsigned grid = (unsigned)((n_waves + kWavesPerWG - 1) / kWavesPerWG);
    if (kLdsBytes > 64 * 1024) {   // once per instantiation and device would do; the call is cheap
        AUR_CHECK_HIP(hipFuncSetAttribute((const void*)stft_frames_kernel<C, LOG, BM, SAMPLES, ZT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    }
    hipLaunchKernelGGL((stft_frames_kernel<C, LOG, BM, SAMPLES, ZT>), dim3(grid), dim3(kWavesPerWG * 64), kLdsBytes, st,
                       d_audio, (long)n_samples, (const long*)d_pos, (long)frame_begin, n_frames, hop, fpw,
                       ctx->d_window, ctx->d_twid, d_logspec, d_flux_sum, d_meanabs, d_samples, d_bandmax);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

// hop <= 896: the zero-tail variant (AUR_STFT_ZT=0 in the environment keeps the general kernel: A/B runs)
template <int C, bool LOG, bool BM, bool SAMPLES>
int launch_stft(aur_ctx* ctx, const float* d_audio, int64_t n_samples, const int64_t* d_pos, int64_t frame_begin,
                int n_frames, int hop, float* d_logspec, float* d_flux_sum, float* d_meanabs, float* d_samples,
                float* d_bandmax, hipStream_t st) {
    if (hop <= AUR_BLOCK - 128 && ctx->stft_zt)
        return launch_stft_z<C, LOG, BM, SAMPLES, true>(ctx, d_audio, n_samples, d_pos, frame_begin, n_frames, hop,
                                                        d_logspec, d_flux_sum, d_meanabs, d_samples, d_bandmax, st);
    return launch_stft_z<C, LOG, BM, SAMPLES, false>(ctx, d_audio, n_samples, d_pos, frame_begin, n_frames, hop,
                                                     d_logspec, d_flux_sum, d_meanabs, d_samples, d_bandmax, st);
}

// Which outputs are wanted is a compile-time property of the kernel (no branches in the frame loop).
template <int C, bool LOG, bool BM>
int launch_stft_s(bool samples, aur_ctx* ctx, const float* a, int64_t n, const int64_t* p, int64_t fb, int nf, int hop,
                  float* ls, float* fs, float* ma, float* sm, float* bm, hipStream_t st) {
    return samples ? launch_stft<C, LOG, BM, true>(ctx, a, n, p, fb, nf, hop, ls, fs, ma, sm, bm, st)
                   : launch_stft<C, LOG, BM, false>(ctx, a, n, p, fb, nf, hop, ls, fs, ma, sm, bm, st);
}
template <int C>
int launch_stft_c(aur_ctx* ctx, const float* a, int64_t n, const int64_t* p, int64_t fb, int nf, int hop, float* ls,
                  float* fs, float* ma, float* sm, float* bm, hipStream_t st) {
    const bool s = sm != nullptr;
    if (ls != nullptr)
        return bm != nullptr ? launch_stft_s<C, true, true>(s, ctx, a, n, p, fb, nf, hop, ls, fs, ma, sm, bm, st)
                             : launch_stft_s<C, true, false>(s, ctx, a, n, p, fb, nf, hop, ls, fs, ma, sm, bm, st);
    return bm != nullptr ? launch_stft_s<C, false, true>(s, ctx, a, n, p, fb, nf, hop, ls, fs, ma, sm, bm, st)
                         : launch_stft_s<C, false, false>(s, ctx, a, n, p, fb, nf, hop, ls, fs, ma, sm, bm, st);
}

}  // namespace

extern "C" int aur_stft_frames(aur_ctx* ctx, const float* d_audio, int64_t n_samples, int channels,
                               const int64_t* d_pos, int64_t frame_begin, int n_frames, int hop, float* d_logspec,
                               float* d_flux_sum, float* d_meanabs, float* d_samples, float* d_bandmax,
                               void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_frames >= 0, "aur_stft_frames: n_frames %d < 0", n_frames);
    if (n_frames == 0) return AUR_OK;
    AUR_REQUIRE(d_audio && d_pos && d_flux_sum && d_meanabs, "aur_stft_frames: null device pointer");
    AUR_REQUIRE(n_samples >= 2, "aur_stft_frames: n_samples %lld < 2", (long long)n_samples);
    AUR_REQUIRE(channels == 1 || channels == 2,
                "aur_stft_frames: channels %d unsupported (1 or 2; mix more channels down with aur_mix_channels first)",
                channels);
    AUR_REQUIRE(hop >= 1, "aur_stft_frames: hop %d < 1", hop);
    AUR_REQUIRE(frame_begin >= 0, "aur_stft_frames: frame_begin %lld < 0", (long long)frame_begin);
    AUR_REQUIRE(((uintptr_t)d_audio & 7u) == 0 &&
                    (d_samples == nullptr || ((uintptr_t)d_samples & 7u) == 0),
                "aur_stft_frames: misaligned device pointer");
    static_assert(kLdsBytes <= 160 * 1024, "LDS of a CU");
    hipStream_t st = (hipStream_t)stream;
    if (channels == 2)
        return launch_stft_c<2>(ctx, d_audio, n_samples, d_pos, frame_begin, n_frames, hop, d_logspec, d_flux_sum,
                                d_meanabs, d_samples, d_bandmax, st);
    return launch_stft_c<1>(ctx, d_audio, n_samples, d_pos, frame_begin, n_frames, hop, d_logspec, d_flux_sum,
                            d_meanabs, d_samples, d_bandmax, st);
}

extern "C" int aur_mix_channels(aur_ctx* ctx, const float* d_audio, int64_t n_samples, int channels, float* d_mono,
                                void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n_samples >= 0, "aur_mix_channels: n_samples %lld < 0", (long long)n_samples);
    AUR_REQUIRE(channels >= 1 && channels <= 64, "aur_mix_channels: channels %d outside 1..64", channels);
    if (n_samples == 0) return AUR_OK;
    AUR_REQUIRE(d_audio && d_mono, "aur_mix_channels: null device pointer");
    AUR_REQUIRE(n_samples < ((int64_t)1 << 39), "aur_mix_channels: n_samples %lld too large", (long long)n_samples);
    hipLaunchKernelGGL(mix_channels_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_audio, (long)n_samples, channels, d_mono);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_stft_buffer(aur_ctx* ctx, const float* d_buffer, float* d_prev_mag, float* d_logspec,
                               float* d_flux_sum, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(d_buffer && d_prev_mag && d_logspec && d_flux_sum, "aur_stft_buffer: null device pointer");
    AUR_REQUIRE(((uintptr_t)d_buffer & 7u) == 0, "aur_stft_buffer: d_buffer must be 8-byte aligned");
    hipLaunchKernelGGL(stft_buffer_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_buffer, d_prev_mag,
                       ctx->d_window, ctx->d_twid, d_logspec, d_flux_sum);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}

extern "C" int aur_log1p_f32(aur_ctx* ctx, const float* d_x, float* d_y, int64_t n, void* stream) {
    int rc = aur_use_device(ctx);
    if (rc) return rc;
    AUR_REQUIRE(n >= 0 && n <= (int64_t)1 << 31, "aur_log1p_f32: n %lld out of range", (long long)n);
    if (n == 0) return AUR_OK;
    AUR_REQUIRE(d_x && d_y, "aur_log1p_f32: null device pointer");
    hipLaunchKernelGGL(log1p_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_x,
                       d_y, (long)n);
    AUR_CHECK_HIP(hipGetLastError());
    return AUR_OK;
}
