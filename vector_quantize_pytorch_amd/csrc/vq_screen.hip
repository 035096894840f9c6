// vq_screen.hip -- screened nearest-code assignment, gfx950 only.  D in {32, 64, 128, 256, 512}; bf16 and fp32 rows; Euclidean
// metric, or cosine on unit-norm rows (METRIC 1).
//
// The exact kernel (vqhip.hip, vq_assign_kernel) evaluates the reference's cdist (vqp.py:58-62) bit for bit on the fp32 MFMA pipe,
// which runs at 1/16 of the 16-bit MFMA rate.  The same INDICES can be had much cheaper, still exactly:
//
//   1. screen: an approximate score t[n, c] ~ x_n . c - ||c||^2 / 2 on the 16-bit MFMA pipe whose error against the reference's own
//      fp32 arithmetic is BOUNDED (eps(n) below); per row the kernel keeps the best, second and third t.
//   2. a row whose margin (best - second) exceeds the bound has a certified winner: every other code is farther in the reference's
//      fp32 arithmetic as well, ties of the rounded sqrt included.  Its index (and gathered code, residual, squared error) are final.
//   3. a row whose margin to the THIRD exceeds the bound has exactly two candidates: two distances in the exact arithmetic decide
//      (vq_pair_kernel, vqhip.hip).  The rest (a fraction of a percent) is re-done by the exact sweep (vq_refine_kernel).
//
// Kernels in this file:
//   vq_screen16_kernel      single-pass fp16 screen, two 32-row blocks per wave, D <= 256: bf16 rows (exact fp16 operands after a
//                           power-of-two scaling) and fp32 rows (one fp16 operand set, measured residual in the bound); paired
//                           sweep (every A fragment read from LDS once for both row blocks, the previous tile's top-3 fold between
//                           this tile's MFMAs); residual chain in the prologue (vqhip_assign_screened_chain)
//   vq_screen16_1rb_kernel  one row block per wave: D = 512, and the two-operand-set fp32 variant (VQHIP_SCREEN_F32_2PART=1)
//   (vq_screen_c.hip: a persistent form of the first one with a cyclic tile stream, opt-in through VQHIP_SCREEN_PERSIST=2.)
// Round 1's two-pass bf16 hi/lo kernels and round 2's flat / skewed sweeps are in the git history (removed in round 3).
//
// Error bound, in units of s = ||x||^2 + ||c||^2 - 2 x.c (u = 2^-24, D features, X = ||x||, Y = max_c ||c||); the kernels state their
// own terms where they compute eps (codebook rounding through the exact residual norm, D + 1 accumulated terms):
//   reference chain (oracle/vq_oracle.c::vqo_assign):  |s_ref - s| <= u (x2 + y2) + u s + 2 D u X Y
//   MFMA accumulation: one rounding per added term with TRUNCATION (2u) -- pessimistic for a fused dot-product unit; measured at
//            < 9 % of the model, tests/test_gpu_screen_fuzz.py::test_mfma_accumulation_error_within_model
//   sqrt collapse: distances that differ by < 4 ulp(s) may round to the same sqrt (vqp.py:62) and tie: 8 u s
//   index bits: the kernel stores the lane-local code number in the 4 low mantissa bits of t: 2 * 16 ulp(t)
// Round 6: the bound is charged PER CODE.  Every term above is a bound for one code's score with Y = ||c||; the codebook-wide maxima
// they used to be evaluated with made one large-norm (or badly rounded) code raise the threshold of every row.  Now
//   e(c) = Rrow + Arow ||c|| + kb ||c||^2,   Rrow = 5 u x2 + X r0 + 2e-8,   Arow = X (u (10 + D + 2.002 (D + 1)) + rho) + ||x' - x_h|| + conv,
// with ||c - c_h|| <= rho ||c|| + r0 (vq_pack16_kernel: rho measured, <= 2^-12).  The sweep ADDS Arow ||c|| + kb ||c||^2 to each code's
// start value (||c|| per code in the tile tail), so what it tracks are upper bounds U_c >= true score - Rrow, and
// A row is certified when  U_best - 2 (Arow ||c_best|| + kb ||c_best||^2) - U_second > 2 Rrow (+ index bits).  tests/test_gpu_ops.py measures the
// actual |t - t_exact| against this bound and tests/test_gpu_screen_fuzz.py checks 1.6e7 adversarial rows per run bit for bit.

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vqhip_internal.h"

#include "vq_screen_args.h"

#ifdef VQ_TRACE
extern long long *vq_g_trace;
#ifndef VQ_TRACE_BLOCK0
#define VQ_TRACE_BLOCK0 0    // first of the 16 traced workgroups (a mid-grid value shows the steady state)
#endif
#define VQ_TB ((int)blockIdx.x - VQ_TRACE_BLOCK0)
#define VQ_STAMP(slot) do { if (a.trace && VQ_TB >= 0 && VQ_TB < 16 && lane == 0 && ct < 64) a.trace[(((size_t)VQ_TB * 4 + wave) * 64 + ct) * 4 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define VQ_PHASE(slot) do { if (a.trace && VQ_TB >= 0 && VQ_TB < 16 && lane == 0) a.trace[16 * 4 * 64 * 4 + ((size_t)VQ_TB * 4 + wave) * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define VQ_STAMP(slot) do {} while (0)
#define VQ_PHASE(slot) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Single-pass fp16 screen for bf16 rows (vq_screen16_kernel).  Half the MFMAs of the hi/lo kernel above:
//   * bf16 rows are converted to fp16 in registers after an exact power-of-two scaling 2^SX chosen per WAVE so that the
//     largest element of the wave's 64 rows lands in [2^14, 2^15): a bf16 value (8 significant bits) is an exact fp16
//     value (11 bits) unless it falls below 2^-14 after scaling, i.e. below 2^-28 of the wave's largest element -- those
//     elements are truncated with an absolute error <= 2^-24 (scaled units) that the threshold charges;
//   * the codebook comes as ONE fp16 part, c_h = fp16(c 2^sc) (vq_pack16_kernel): 11 significant bits instead of the
//     16 of the hi/lo split.  The certificate charges the rounding through the exact residual norm,
//     |x.(c - c_h)| <= X * max_c ||c - c_h|| (scalars[1]), for both codes of the margin;
//   * scores live in scaled units, t' = 2^(SX + sc) (x.c_h - ||c||^2 / 2): the accumulator starts at -||c||^2/2 * 2^(SX+sc)
//     (one v_mul per score register and tile, shared by both row blocks through the MFMA's separate C operand);
//   * the MFMA accumulation term of the bound halves as well (D + 1 added terms instead of 2 D + 1);
//   * rows that fail the (larger) threshold go to the exact pass as before.  Measured on the reference's default init at
//     cfg 2: see DESIGN.md.
// The loop processes SUB 32-code tiles per barrier (one LDS buffer = SUB tiles, double buffered).
// Outputs (q rows, residual rows, squared error) are produced by a row-cooperative pass that re-reads x -- the registers
// hold the scaled fp16 copy -- which also frees the sweep from the output phase's registers.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#ifndef VQS16_PF
#define VQS16_PF 4
#endif

template <int DT> struct Screen16Cfg {
    static constexpr int TILE_B = 64 * DT + 1024;
#ifdef VQS16_SUB
    static constexpr int SUB = VQS16_SUB;
#else
    static constexpr int SUB = DT >= 512 ? 1 : (DT == 256 ? 2 : (DT == 128 ? 4 : 8));   // 32-code tiles per barrier (divides VQ_F16_TILE_GROUP)
#endif
    static constexpr int BUF_B = (SUB * TILE_B / 1024 + VQS_WAVES - 1) / VQS_WAVES * VQS_WAVES * 1024;   // whole pieces for every wave
    static constexpr int SMEM = 2 * BUF_B;
};

// XF32: fp32 rows with ONE fp16 operand set per row, x_h = fp16_rne(x 2^SX) -- the same sweep as for bf16 rows (two row blocks per
// wave, half the MFMAs and LDS reads per row of the two-set kernel further down).  The certificate charges the MEASURED residual
// ||x' - x_h|| Y per code.  The scale is chosen per row block (rows of different blocks are never compared), so that a row block's
// raw fp32 values (128 registers at D = 256) are converted before the next one is loaded.
// NPART = 2 (fp32 rows, D <= 128, where two operand sets per row block still fit the registers): x' = x_h + x_m, both truncated
// fp16 parts (|x' - x_h - x_m| <= 2^-20 |x'|), two MFMAs per k-step on one A fragment -- the x side then costs the certificate
// 2^-20 X Y instead of the measured 2^-11-level residual, which brings the uncertified fraction of fp32 rows down to bf16 levels.
template <int DT, int METRIC, bool XF32 = false, int NPART = 1>
__global__ void __launch_bounds__(VQS_WAVES * 64, 8 / VQS_WAVES) vq_screen16_kernel(const ScreenArgs a0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ScreenArgs a = vq_head_screen_args(a0);
    using Cfg = Screen16Cfg<DT>;
    constexpr int TILE_B = Cfg::TILE_B;
    constexpr int SUB = Cfg::SUB;
    constexpr int SUPER_B = SUB * TILE_B;           // bytes per LDS buffer
    constexpr int NCHUNK = SUPER_B / 1024;          // 1-KiB pieces per buffer
    constexpr int NK = DT / 16;                     // MFMA k-steps = steps per tile (one MFMA per row block each)
    constexpr int PMAX = (NCHUNK + VQS_WAVES - 1) / VQS_WAVES;   // pieces per wave and buffer
    constexpr int BUF_B = Cfg::BUF_B;               // LDS bytes per buffer: PMAX pieces for EVERY wave, so that no copy is conditional
    constexpr int PPS = (PMAX + SUB - 1) / SUB;     // pieces a wave copies during one tile
    constexpr int NBS = (NK >= 16) ? 2 : 1;         // staging batches per tile
    [[maybe_unused]] constexpr int BS = (PPS + NBS - 1) / NBS;
    [[maybe_unused]] constexpr int SPAN = NK / NBS;                  // steps between batch starts
    [[maybe_unused]] constexpr int LAG = (SPAN >= 8) ? SPAN - 2 : SPAN - 1;   // steps between a batch's loads and its LDS stores
    static_assert(VQ_F16_TILE_GROUP % SUB == 0, "tile padding must cover the tiles of one barrier");
    static_assert(LAG >= 1 && (NBS - 1) * SPAN + LAG < NK, "staging schedule");
    [[maybe_unused]] constexpr int BS2 = (PPS + 1) / 2;              // skewed sweep: two staging batches per tile, one per row-block phase
#ifdef VQS16_LAG2
    [[maybe_unused]] constexpr int LAG2 = (NK > VQS16_LAG2 + 1) ? VQS16_LAG2 : NK - 1;
#else
    [[maybe_unused]] constexpr int LAG2 = (NK >= 8) ? NK - 2 : NK - 1;
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int half = lane >> 5;
    const int64_t wrow0 = (int64_t)blockIdx.x * VQ_SCREEN_ROWS + wave * 64;
    VQ_PHASE(0);
#ifdef VQ_DEV_SWITCHES
    if (a.stagger > 0 && (int)blockIdx.x < a.stagger_first) {
        // the workgroup whose LDS allocation does not start at 0 is the second one on its CU (HW_REG_LDS_ALLOC, LDS_BASE field)
        const bool second_wg = (__builtin_amdgcn_s_getreg((11 << 11) | (0 << 6) | 6) & 0xfff) != 0;
        if (second_wg) {
            const long long wait = 1024ll * a.stagger, t0 = __builtin_readcyclecounter();
            while ((long long)__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(8);
        }
    }
#endif

    // ---- first buffer: wave w copies the 1-KiB pieces w, w + WAVES, ... ----
    constexpr int PSTRIDE = VQS_WAVES * 1024;
    const int piece_off = wave * 1024 + lane * 16;
    const char *tiles = a.tiles16;

    // ---- x rows, requested first: lane (j, half) holds x[row][16 ks + 8 half + 0..7] for every k-step ----
    uint4 xb[2][NK];
    uint4 xm[2][NPART == 2 ? NK : 1];   // second operand set (NPART == 2)
    int64_t rows[2];
    static_assert(NPART == 1 || XF32, "only fp32 rows need a second operand set");
    bool row_ok[2];
    float xs2[2], rxn[2] = {0.f, 0.f};     // ||x||^2 (x 1.001) and, for fp32 rows, ||x' - x_h|| in unscaled units
    [[maybe_unused]] float xs_wave = 0.f;  // bf16 rows: the largest finite ||x||^2 of the wave's 64 rows
    int SXv[2];
    const int sc = (int)a.scalars[2];
    // scale exponent from the largest FINITE squared row norm `mx` (float bits) of the rows sharing it.  Every element is <= ||x||, so
    // bringing that norm below 2^14 keeps every element below fp16's 65504 (rows with a non-finite norm can never be certified
    // anyway); elements then sit around 2^14 / sqrt(D), far above fp16's 2^-14.  sc is the codebook's (scalars[2]); SX + sc stays
    // inside fp32's exponent range so that every power of two below is exact, and SX - sc <= 90 keeps -||c||^2/2 * 2^(SX+sc) finite
    // (rows that tiny against the codebook lose bits in the conversion, which `conv` charges).
    auto pick_sx = [&](unsigned mx) {
        const int e2 = (int)(mx >> 23) - 127;          // largest ||x||^2 in [2^e2, 2^(e2+1))  =>  ||x|| < 2^((e2 >> 1) + 1)
        int SX = (mx == 0u) ? 0 : 14 - ((e2 >> 1) + 1);
        SX = SX > 120 - sc ? 120 - sc : SX;
        SX = SX < -120 - sc ? -120 - sc : SX;
        SX = SX > sc + 90 ? sc + 90 : SX;
        return SX > 126 ? 126 : (SX < -126 ? -126 : SX);
    };
    auto wave_max_finite = [&](float v) {
        const unsigned bits = __float_as_uint(v);
        unsigned mx = (bits & 0x7f800000u) == 0x7f800000u ? 0u : (bits & 0x7fffffffu);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)mx, o, 64); mx = t > mx ? t : mx; }
        return (unsigned)__builtin_amdgcn_readfirstlane((int)mx);
    };
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        rows[rb] = wrow0 + rb * 32 + j;
        row_ok[rb] = rows[rb] < a.N;
    }
    if (!XF32) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int64_t rc = row_ok[rb] ? rows[rb] : (a.N - 1);
            const unsigned short *p = (const unsigned short *)a.x + rc * a.ldx + 8 * half;
#pragma unroll
#ifdef VQS16_NO_XLOAD      // A/B: no row traffic in the prologue (one 16-byte piece per lane, replicated)
            for (int ks = 0; ks < NK; ++ks) xb[rb][ks] = *(const uint4 *)(p);
#else
            for (int ks = 0; ks < NK; ++ks) xb[rb][ks] = *(const uint4 *)(p + ks * 16);
#endif
        }
    }
    // ---- first buffer: wave w copies the 1-KiB pieces w, w + WAVES, ... (PMAX of them: a piece past the buffer's end reads
    //      the next tiles / the tail pad and lands in the LDS pad) ----
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        *(f32x4 *)(smem + piece_off + k * PSTRIDE) = *(const f32x4 *)(tiles + piece_off + (size_t)k * PSTRIDE);

    if (!XF32) {
        // ---- bf16 rows: ||x||^2 per row (any order: it only scales the bound), one scale for the wave, exact conversion ----
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float xs = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const unsigned w[4] = {xb[rb][ks].x, xb[rb][ks].y, xb[rb][ks].z, xb[rb][ks].w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[q]), __builtin_bit_cast(bf16x2, w[q]), xs, false);
            }
            xs += __shfl_xor(xs, 32, 64);
            xs2[rb] = xs * 1.001f;
        }
        const unsigned m0 = wave_max_finite(xs2[0]), m1b = wave_max_finite(xs2[1]);
        xs_wave = __uint_as_float(m0 > m1b ? m0 : m1b);
        SXv[0] = SXv[1] = pick_sx(m0 > m1b ? m0 : m1b);
        const float S = __uint_as_float((unsigned)(SXv[0] + 127) << 23);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                unsigned w[4] = {xb[rb][ks].x, xb[rb][ks].y, xb[rb][ks].z, xb[rb][ks].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = __uint_as_float(w[q] << 16) * S, hi = __uint_as_float(w[q] & 0xffff0000u) * S;
                    w[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));   // exact above 2^-14, truncated below (conv)
                }
                xb[rb][ks] = make_uint4(w[0], w[1], w[2], w[3]);
            }
    } else {
        // ---- fp32 rows, one row block at a time: raw values -> ||x||^2 -> the block's scale -> x_h = fp16_rne(x') and the residual ----
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifndef VQS16_F32_DIRECT
        if constexpr (NPART == 1) {
            // Whole rows per wave instruction (LPR lanes x 16 bytes per row; one row per lane, the operand layout's pattern, touches
            // 64 cache lines per instruction and ran this load at 1.6 TB/s).  Norms, scale, conversion and the measured residual
            // all happen in that row-cooperative layout; only the fp16 operands are transposed into the operand layout, through
            // this wave's quarter of the SECOND codebook buffer, which nothing stages before the barrier that ends the prologue.
            constexpr int CH = DT < 64 ? DT : 64, LPR = CH / 4, RPI = 64 / LPR, NI = 32 / RPI, NSTEP = DT / CH;
            constexpr int COLS = DT < 128 ? DT : 128, NPASS = DT / COLS, SPP = COLS / CH, PITCH = COLS * 2 + 16;
            static_assert(32 * PITCH + 256 <= Cfg::BUF_B / VQS_WAVES, "transposition scratch");
            char *scr = smem + Cfg::BUF_B + wave * (Cfg::BUF_B / VQS_WAVES);
            float *nrm = (float *)(scr + 32 * PITCH);          // [2][32]: ||x||^2 and ||x' - x_h||^2 per row of the block
            const int lr = lane / LPR, lc = lane % LPR;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                f32x4 g[NSTEP][NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int64_t r = wrow0 + rb * 32 + i * RPI + lr;
                    const float *pr = (const float *)a.x + (r < a.N ? r : a.N - 1) * a.ldx + lc * 4;
#pragma unroll
                    for (int st = 0; st < NSTEP; ++st) g[st][i] = *(const f32x4 *)(pr + st * CH);
                }
                if (a.prev_idx) {
                    // residual chain (rvq.py:524: residual = residual - quantized.detach()): the loaded rows are the PREVIOUS stage's
                    // input; its code rows (fp32, from L2) are subtracted here -- the same fp32 x - q the output phase of that stage
                    // would have written -- and the result is both this stage's input and, stored to x_out, the tensor the exact
                    // passes and the statistics pass of this stage read.  Saves the previous stage's re-read of x for the residual.
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int64_t r = wrow0 + rb * 32 + i * RPI + lr;
                        const int64_t rc = r < a.N ? r : a.N - 1;
                        const float *pe = a.prev_embed + (size_t)a.prev_idx[rc * a.prev_idx_stride] * DT + lc * 4;
                        f32x4 e[NSTEP];
#pragma unroll
                        for (int st = 0; st < NSTEP; ++st) e[st] = *(const f32x4 *)(pe + st * CH);
#pragma unroll
                        for (int st = 0; st < NSTEP; ++st) {
                            g[st][i] = g[st][i] - e[st];
                            if (r < a.N && a.x_out) *(f32x4 *)(a.x_out + r * a.ldxo + lc * 4 + st * CH) = g[st][i];
                        }
                    }
                }
                float ps[NI];
                unsigned mxb = 0u;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    float xs = 0.f;
#pragma unroll
                    for (int st = 0; st < NSTEP; ++st) {
                        const f32x4 v = g[st][i];
                        xs = __builtin_fmaf(v.x, v.x, xs); xs = __builtin_fmaf(v.y, v.y, xs);
                        xs = __builtin_fmaf(v.z, v.z, xs); xs = __builtin_fmaf(v.w, v.w, xs);
                    }
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) xs += __shfl_xor(xs, o, 64);
                    ps[i] = xs * 1.001f;
                    const unsigned bits = __float_as_uint(ps[i]);
                    const unsigned fin = (bits & 0x7f800000u) == 0x7f800000u ? 0u : (bits & 0x7fffffffu);
                    mxb = fin > mxb ? fin : mxb;
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)mxb, o, 64); mxb = t > mxb ? t : mxb; }
                SXv[rb] = pick_sx((unsigned)__builtin_amdgcn_readfirstlane((int)mxb));
                const float S = __uint_as_float((unsigned)(SXv[rb] + 127) << 23);
                uint2 hq[NSTEP][NI];
                float pr2[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    float r2 = 0.f;
#pragma unroll
                    for (int st = 0; st < NSTEP; ++st) {
                        const float v[4] = {g[st][i].x * S, g[st][i].y * S, g[st][i].z * S, g[st][i].w * S};
                        unsigned hw[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            f16x2 h;
                            h[0] = (_Float16)v[2 * q]; h[1] = (_Float16)v[2 * q + 1];
                            const float r0 = v[2 * q] - (float)h[0], r1 = v[2 * q + 1] - (float)h[1];   // exact
                            r2 = __builtin_fmaf(r0, r0, r2); r2 = __builtin_fmaf(r1, r1, r2);
                            hw[q] = __builtin_bit_cast(unsigned, h);
                        }
                        hq[st][i] = make_uint2(hw[0], hw[1]);
                    }
#pragma unroll
                    for (int o = 1; o < LPR; o <<= 1) r2 += __shfl_xor(r2, o, 64);
                    pr2[i] = r2;
                }
                if (lc == 0) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) { nrm[i * RPI + lr] = ps[i]; nrm[32 + i * RPI + lr] = pr2[i]; }
                }
#pragma unroll
                for (int ps_ = 0; ps_ < NPASS; ++ps_) {
#pragma unroll
                    for (int sl = 0; sl < SPP; ++sl)
#pragma unroll
                        for (int i = 0; i < NI; ++i)
                            *(uint2 *)(scr + (i * RPI + lr) * PITCH + (sl * CH + lc * 4) * 2) = hq[ps_ * SPP + sl][i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int kk = 0; kk < COLS / 16; ++kk)
                        xb[rb][ps_ * (COLS / 16) + kk] = *(const uint4 *)(scr + j * PITCH + (16 * kk + 8 * half) * 2);
                    if (ps_ == NPASS - 1) {
                        xs2[rb] = nrm[j];
                        rxn[rb] = sqrtf(nrm[32 + j] * 1.001f) * 1.001f * __uint_as_float((unsigned)(127 - SXv[rb]) << 23);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        } else
#endif
        {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x4 xr[NK][2];
            const float *p = (const float *)a.x + (row_ok[rb] ? rows[rb] : (a.N - 1)) * a.ldx + 8 * half;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) { xr[ks][0] = *(const f32x4 *)(p + ks * 16); xr[ks][1] = *(const f32x4 *)(p + ks * 16 + 4); }
            float xs = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 v = xr[ks][q];
                    xs = __builtin_fmaf(v.x, v.x, xs); xs = __builtin_fmaf(v.y, v.y, xs);
                    xs = __builtin_fmaf(v.z, v.z, xs); xs = __builtin_fmaf(v.w, v.w, xs);
                }
            xs += __shfl_xor(xs, 32, 64);
            xs2[rb] = xs * 1.001f;
            SXv[rb] = pick_sx(wave_max_finite(xs2[rb]));
            const float S = __uint_as_float((unsigned)(SXv[rb] + 127) << 23);
            float r2 = 0.f;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const float v[8] = {xr[ks][0].x * S, xr[ks][0].y * S, xr[ks][0].z * S, xr[ks][0].w * S,
                                    xr[ks][1].x * S, xr[ks][1].y * S, xr[ks][1].z * S, xr[ks][1].w * S};
                unsigned hw[4], mw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (NPART == 1) {
                        f16x2 h;
                        h[0] = (_Float16)v[2 * q]; h[1] = (_Float16)v[2 * q + 1];
                        const float r0 = v[2 * q] - (float)h[0], r1 = v[2 * q + 1] - (float)h[1];   // exact
                        r2 = __builtin_fmaf(r0, r0, r2); r2 = __builtin_fmaf(r1, r1, r2);
                        hw[q] = __builtin_bit_cast(unsigned, h);
                    } else {               // x_h: truncated x', x_m: truncated exact remainder
                        const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[2 * q], v[2 * q + 1]));
                        const float r0 = v[2 * q] - (float)h[0], r1 = v[2 * q + 1] - (float)h[1];
                        hw[q] = __builtin_bit_cast(unsigned, h);
                        mw[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
                    }
                }
                xb[rb][ks] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                if (NPART == 2) xm[rb][ks] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
            }
            r2 += __shfl_xor(r2, 32, 64);
            rxn[rb] = sqrtf(r2 * 1.001f) * 1.001f * __uint_as_float((unsigned)(127 - SXv[rb]) << 23);
            __builtin_amdgcn_sched_barrier(0);   // finish this block before the next block's 128 raw registers are requested
        }
    }
    }
    float SSv[2], iSSv[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        SSv[rb] = __uint_as_float((unsigned)(SXv[rb] + sc + 127) << 23);
        iSSv[rb] = __uint_as_float((unsigned)(127 - SXv[rb] - sc) << 23);
    }

    // ---- the certificate's pieces (unscaled units; file header).  Per code c the screen's score errs against the reference's own
    //      arithmetic by at most  e(c) = Rrow + Arow ||c|| + kb ||c||^2  (kb ||c||^2 is inside the tile's start value, vq_pack16_kernel):
    //      the sweep adds Arow ||c|| to every start value, so the tracked scores are UPPER bounds U_c = t_c + (code part of e(c)), and
    //      the winner is certified when  U_1 - 2 (code part of the winner) - U_2 > 2 Rrow  -- no codebook-wide maximum anywhere ----
    float ArowS[2], Rrow[2], Arow[2];
    {
        const float rho = __uint_as_float(a.scalars[1]);
        const float r0 = __uint_as_float(a.scalars[3]);
        const float u = 5.9604645e-8f;   // 2^-24
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float iS = __uint_as_float((unsigned)(127 - SXv[rb]) << 23);
            const float xs = xs2[rb];
            const float xn = sqrtf(xs) * 1.0001f;
            // (per ROW: a factor built from the wave's largest row norm -- one start vector for both row blocks -- was measured: it saves
            //  ~1 % of the cfg-2 search and sends 55 % instead of 7 % of the rows of a batch whose row norms span 30 x to the exact sweep)
            const float xa = xn;
            const float nacc = (float)(NPART * DT + 1);
            // truncated elements: |dx_k| <= 2^-24 / S each, sum_k |dx_k| |c_k| <= 2^-24 / S * sqrt(D) * ||c||   (per operand set)
            const float conv = NPART * u * sqrtf((float)DT) * iS;
            // the part of x' the operand set(s) do not carry: one set -- the measured ||x' - x_h|| ||c||; two sets -- 2^-20 X ||c||
            const float drop = !XF32 ? 0.f : (NPART == 1 ? rxn[rb] : 9.5367432e-7f * 1.01f * xn);
            if (METRIC == 0) {
                Arow[rb] = (xa * (u * (10.f + (float)DT + 2.002f * nacc) + rho) + drop + conv) * (1.0001f + 4.f * nacc * u);   // (+ the MFMA's rounding of the added part itself)
                Rrow[rb] = 5.f * u * xs + xn * r0 + 2e-8f;
            } else {
                Arow[rb] = (xa * (u * (float)(DT + 2 * NPART * DT) * 1.001f + rho) + drop + conv) * (1.0001f + 4.f * nacc * u);
                Rrow[rb] = xn * r0 + 1e-30f;
            }
            ArowS[rb] = Arow[rb] * SSv[rb];
            // overflow guard: a row norm or a code norm beyond fp32 (or a scaled factor that left its range) -> this row is never
            // certified (Rrow = inf) and adds nothing to the start values (no inf x 0 = NaN may enter the fold)
            const bool y2ok = __uint_as_float(a.scalars[0]) < 1e37f;
            const bool ok = (Arow[rb] < 1e30f) & (ArowS[rb] < 1e30f) & (Rrow[rb] < 1e30f) & y2ok;      // (selects: a branch here made hipcc spill)
            Arow[rb] = ok ? Arow[rb] : 0.f; ArowS[rb] = ok ? ArowS[rb] : 0.f; Rrow[rb] = ok ? Rrow[rb] : __builtin_inff();
        }
    }

    float m1[2] = {-__builtin_inff(), -__builtin_inff()};
    float m2[2] = {-__builtin_inff(), -__builtin_inff()};
    float m3[2] = {-__builtin_inff(), -__builtin_inff()};
    int tix[2] = {0, 0}, tix2[2] = {0, 0};
    VQ_PHASE(1);

#ifdef VQS16_NO_SWEEP      // A/B: prologue and output phases only
    const int nst = 0;
#else
    const int nst = a.n_tiles16 / SUB;   // barriers
#endif
    // ---- paired sweep: every A fragment is read from LDS ONCE and multiplied with both row blocks (two MFMAs on DIFFERENT
    //      accumulators back to back: no dependent-accumulator stall between them), and the top-3 epilogue of the PREVIOUS tile's
    //      two accumulators is issued in slices between this tile's MFMAs.  Tiles are processed in pairs with the accumulator
    //      sets swapping roles (A accumulates while B is folded, then B accumulates while A is folded), so no accumulator is
    //      ever copied.  Against the skewed sweep below: half the LDS reads, and a wave's MFMA stream alternates accumulators
    //      (a same-accumulator chain with VALU between its MFMAs pays the write-back latency at every step). ----
    static_assert(SUB % 2 == 0, "tiles are swept in pairs");
#ifndef VQS16_PF2
#define VQS16_PF2 3
#endif
    constexpr int PF2 = VQS16_PF2 < NK ? VQS16_PF2 : NK;
    constexpr int FPS = 16 / NK;                        // folds per accumulator and step
    static_assert(FPS * NK == 16, "16 scores per accumulator spread evenly over the k-steps");
    constexpr bool TWO_B = NK >= 8;                     // staging batches per tile
    constexpr int BSF = TWO_B ? (PPS + 1) / 2 : PPS;
    constexpr int LAGF = TWO_B ? NK / 2 - 1 : NK - 1;   // steps (of two MFMAs) between a batch's loads and its LDS stores
    auto fold = [&](const f32x16 &acc, int e, float &b1, float &b2, float &b3) __attribute__((always_inline)) {
        const float k = __uint_as_float((__float_as_uint(acc[e]) & 0xfffffff0u) | (unsigned)e);
        asm volatile("v_med3_f32 %2, %1, %2, %3\n\tv_med3_f32 %1, %0, %1, %3\n\tv_max_f32 %0, %0, %3"
                     : "+v"(b1), "+v"(b2), "+v"(b3) : "v"(k));
    };
    auto book = [&](float om1, float om2, float n1, float n2, int &t1, int &t2, int tile_id) __attribute__((always_inline)) {
        const bool c1 = n1 != om1;
        const int from_old_best = (c1 && n2 == om1) ? t1 : tile_id;
        t2 = (n2 != om2) ? from_old_best : t2;
        t1 = c1 ? tile_id : t1;
    };
    int ptile = 0;
    // one tile: accumulate into (C0, C1), fold the previous tile's (P0, P1).  lane16 = 16 * lane, re-derived once per barrier
    // interval (two instructions) instead of living in a register across the sweep: every lane-dependent address of the loop
    // (A fragments, start values, staging source and destination) is that one value plus something wave-uniform.
    auto tile_body = [&](const char *sbase, int sub, int st, const char *gsrc, char *ldst, unsigned lane16,
                         f32x16 &C0, f32x16 &C1, const f32x16 &P0, const f32x16 &P1) __attribute__((always_inline)) {
        const char *tile = sbase + sub * TILE_B;
        const float *nh = (const float *)(tile + 64 * DT + ((lane16 >> 9) << 4)) ;   // + 4 * half floats
        const int tile_id = st * SUB + sub;
        const bool has_pad = (tile_id + 1) * 32 > a.C;
        const uint4 *ap = (const uint4 *)(tile + lane16);
        f32x4 stg[BSF];
        const int p0 = sub * PPS;
        const float o10 = m1[0], o20 = m2[0], o11 = m1[1], o21 = m2[1];
        uint4 af[PF2];
#pragma unroll
        for (int p = 0; p < PF2; ++p) af[p] = ap[p * 64];
        // start value -||c||^2 / 2 (scaled); register e <-> code 8 (e >> 2) + 4 half + (e & 3) of the tile.  Tiles with padding
        // codes clamp it to a finite -3e38 (see the skewed sweep).  fp32 rows: each row block has its own scale.
        // + (row factor) x ||c|| (tile tail, floats 32 ..): the score becomes an upper bound of the code's true score (certificate above)
        f32x16 init0, init1;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *(const f32x4 *)(nh + 8 * q);
                const f32x4 w = *(const f32x4 *)(nh + 32 + 8 * q);
                if (METRIC != 0) { v.x = v.x < -1e38f ? v.x : 0.f; v.y = v.y < -1e38f ? v.y : 0.f;
                                   v.z = v.z < -1e38f ? v.z : 0.f; v.w = v.w < -1e38f ? v.w : 0.f; }
                // two scores per instruction (v_pk_mul_f32 / v_pk_fma_f32: written as 2-vectors, the SLP vectoriser is off -- csrc/Makefile)
                const f32x2 vp[2] = {f32x2{v.x, v.y}, f32x2{v.z, v.w}}, wp[2] = {f32x2{w.x, w.y}, f32x2{w.z, w.w}};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x2 r0, r1;
                    if (METRIC == 0 || has_pad) {
                        r0 = __builtin_elementwise_fma(wp[i], f32x2{ArowS[0], ArowS[0]}, vp[i] * f32x2{SSv[0], SSv[0]});
                        r1 = __builtin_elementwise_fma(wp[i], f32x2{ArowS[1], ArowS[1]}, vp[i] * f32x2{SSv[1], SSv[1]});
                    } else {
                        r0 = wp[i] * f32x2{ArowS[0], ArowS[0]};
                        r1 = wp[i] * f32x2{ArowS[1], ArowS[1]};
                    }
                    init0[4 * q + 2 * i] = r0.x; init0[4 * q + 2 * i + 1] = r0.y;
                    init1[4 * q + 2 * i] = r1.x; init1[4 * q + 2 * i + 1] = r1.y;
                }
            }
            if (has_pad) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { init0[r] = fmaxf(init0[r], -3.0e38f); init1[r] = fmaxf(init1[r], -3.0e38f); }
            }
        }
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            const f16x8 av = __builtin_bit_cast(f16x8, af[s % PF2]);
            if (s == 0) {
                C0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, xb[0][s]), init0, 0, 0, 0);
                C1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, xb[1][s]), init1, 0, 0, 0);
            } else {
                C0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, xb[0][s]), C0, 0, 0, 0);
                C1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, xb[1][s]), C1, 0, 0, 0);
            }
            if (NPART == 2) {   // the low part of the rows against the same A fragment
                C0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, xm[0][NPART == 2 ? s : 0]), C0, 0, 0, 0);
                C1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, xm[1][NPART == 2 ? s : 0]), C1, 0, 0, 0);
            }
            if (s + PF2 < NK) af[s % PF2] = ap[(s + PF2) * 64];
#ifndef VQS16_NO_EPI
#pragma unroll
            for (int f = 0; f < FPS; ++f) {
                fold(P0, s * FPS + f, m1[0], m2[0], m3[0]);
                fold(P1, s * FPS + f, m1[1], m2[1], m3[1]);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);   // pins the slice and the prefetch distance between the MFMAs
#ifndef VQS16_NO_STAGE
#pragma unroll
            for (int bt = 0; bt < (TWO_B ? 2 : 1); ++bt) {
                const int s0 = bt * (NK / 2);
                if (s == s0) {
#pragma unroll
                    for (int i = 0; i < BSF; ++i)   // unconditional: a piece past this wave's share repeats its last one
                        if (bt * BSF + i < PPS) {
                            const int pc = min(p0 + bt * BSF + i, PMAX - 1);
                            stg[i] = *(const f32x4 *)(gsrc + (size_t)pc * PSTRIDE + lane16);   // uniform base + 32-bit lane offset
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s == s0 + LAGF) {
#pragma unroll
                    for (int i = 0; i < BSF; ++i)
                        if (bt * BSF + i < PPS) *(f32x4 *)(ldst + min(p0 + bt * BSF + i, PMAX - 1) * PSTRIDE + lane16) = stg[i];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#endif
        }
#ifndef VQS16_NO_EPI
        book(o10, o20, m1[0], m2[0], tix[0], tix2[0], ptile);
        book(o11, o21, m1[1], m2[1], tix[1], tix2[1], ptile);
#else
        m1[0] = fmaxf(m1[0], P0[0]); m1[1] = fmaxf(m1[1], P1[0]);
#endif
        ptile = tile_id;
    };
    f32x16 accA0, accA1, accB0, accB1;
#pragma unroll
    for (int r = 0; r < 16; ++r) accB0[r] = accB1[r] = -3.0e38f;      // "padding codes": never win against a real one
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        [[maybe_unused]] const int ct = st;   // trace index
        VQ_STAMP(0);
        __syncthreads();     // buffer `buf` has landed for every wave; the other buffer is free
        VQ_STAMP(1);
        const char *sbase = smem + buf * BUF_B;
        const bool more = st + 1 < nst;
        const char *gsrc = tiles + (size_t)(more ? st + 1 : st) * SUPER_B + wave * 1024;   // wave-uniform part of the source address
        char *ldst = smem + (buf ^ 1) * BUF_B + wave * 1024;   // (the last interval re-copies its own buffer into the idle one: no branch)
        unsigned lane16;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(lane16));
#pragma unroll 1
        for (int sub = 0; sub < SUB; sub += 2) {
            tile_body(sbase, sub, st, gsrc, ldst, lane16, accA0, accA1, accB0, accB1);
            tile_body(sbase, sub + 1, st, gsrc, ldst, lane16, accB0, accB1, accA0, accA1);
        }
        VQ_STAMP(2);
    }
    {   // both row blocks of the last tile
        const float o10 = m1[0], o20 = m2[0], o11 = m1[1], o21 = m2[1];
#ifndef VQS16_NO_EPI
#pragma unroll
        for (int e = 0; e < 16; ++e) { fold(accB0, e, m1[0], m2[0], m3[0]); fold(accB1, e, m1[1], m2[1], m3[1]); }
#endif
        book(o10, o20, m1[0], m2[0], tix[0], tix2[0], ptile);
        book(o11, o21, m1[1], m2[1], tix[1], tix2[1], ptile);
    }

    VQ_PHASE(2);   // sweep done
    // ---- merge the half-waves (each holds the top 3 of its 16 of a tile's 32 codes), classify, emit ----
    //   certified:  best - second > thr                      -> final here
    //   pair:       best - third  > thr (second is too close) -> vq_pair_kernel decides between the two codes exactly
    //   open:       otherwise                                 -> full exact sweep (vq_refine_kernel)
    int code[2], cls[2], id2s[2];
    bool flagged[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int e1 = (int)(__float_as_uint(m1[rb]) & 15u), e2 = (int)(__float_as_uint(m2[rb]) & 15u);
        const int ia1 = tix[rb] * 32 + 8 * (e1 >> 2) + 4 * half + (e1 & 3);
        const int ia2 = tix2[rb] * 32 + 8 * (e2 >> 2) + 4 * half + (e2 & 3);
        const float a1 = m1[rb], a2 = m2[rb], a3 = m3[rb];
        const float p1 = __shfl_xor(a1, 32, 64), p2 = __shfl_xor(a2, 32, 64), p3 = __shfl_xor(a3, 32, 64);
        const int ib1 = __shfl_xor(ia1, 32, 64), ib2 = __shfl_xor(ia2, 32, 64);
        // merge of two descending triples (a from this half, p from the partner)
        const bool take = p1 > a1;
        const float h1 = take ? p1 : a1, h2 = take ? p2 : a2, h3 = take ? p3 : a3;   // the triple that holds the best
        const float l1 = take ? a1 : p1, l2 = take ? a2 : p2;                          // the other one
        const int ih1 = take ? ib1 : ia1, ih2 = take ? ib2 : ia2, il1 = take ? ia1 : ib1;
        const float b1 = h1;
        const bool second_low = l1 > h2;                  // runner-up comes from the other triple
        const float b2 = second_low ? l1 : h2;
        const int id2 = second_low ? il1 : ih2;
        const float b3 = second_low ? fmaxf(h2, l2) : fmaxf(h3, l1);
        code[rb] = ih1;
        // the winner's own code part, A ||c|| + kb ||c||^2 (what the sweep added to its score, and the bound of its error): ||c|| from
        // the tile tail of the packed codebook
        const float kb = 5.9604645e-8f * (5.f + 1.001f * (float)(DT + 1) + 0.51f) * 1.001f;
        auto code_part = [&](int c) {
            const int cc = c < a.C ? c : 0;
            const float yb = *(const float *)(a.tiles16 + (size_t)(cc >> 5) * TILE_B + 64 * DT + (32 + (cc & 31)) * 4);
            return (Arow[rb] * yb + (METRIC == 0 ? kb * yb * yb * 1.001f : 0.f)) * 1.0001f;
        };
        const float cp1 = code_part(ih1);
        const float thr = (2.f * cp1 + 2.f * Rrow[rb]) * SSv[rb] + 8e-6f * fabsf(b1);
#ifdef VQS16_NO_SWEEP
        const bool certified = true;
#else
        const bool certified = ((b1 - b2) > thr) && code[rb] < a.C && b1 < 3.0e38f;
#endif
        const bool pair = !certified && ((b1 - b3) > thr) && code[rb] < a.C && id2 < a.C && b1 < 3.0e38f;
        flagged[rb] = !certified;
        if (row_ok[rb] && half == 0) {
            a.idx_out[rows[rb] * a.idx_stride] = (int64_t)(code[rb] < a.C ? code[rb] : 0);
            if (a.dbg) {
                // debug view in the units of t = x.c - ||c||^2 / 2: the two best scores with exactly what the sweep added to their start
                // values taken off again (the start value is recomputed with the sweep's own operations; the plain -||c||^2 / 2 sits at
                // floats 64 .. of the tile tail), and the margin they have to exceed (certified <=> d[0] - d[1] > d[2])
                float *d = a.dbg + rows[rb] * 4;
                float add[2], sc2[2];
                const int cs[2] = {ih1, id2};
                const float us[2] = {b1, b2};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int cc = cs[k] < a.C ? cs[k] : 0;
                    const float *tl = (const float *)(a.tiles16 + (size_t)(cc >> 5) * TILE_B + 64 * DT) + (cc & 31);
                    const float v = METRIC == 0 ? tl[0] : 0.f, w = tl[32], pz = METRIC == 0 ? tl[64] : 0.f;
                    const float init = METRIC == 0 ? __builtin_fmaf(w, ArowS[rb], v * SSv[rb]) : w * ArowS[rb];
                    sc2[k] = (us[k] - init) * iSSv[rb] + pz;
                    add[k] = init * iSSv[rb] - pz;
                }
                d[0] = sc2[0]; d[1] = sc2[1]; d[2] = thr * iSSv[rb] - add[0] + add[1]; d[3] = certified ? 0.f : (pair ? 2.f : 1.f);
            }
        }
        if (code[rb] >= a.C) code[rb] = 0;
        const bool on = row_ok[rb] && half == 0;
        cls[rb] = !on ? 0 : (certified ? 0 : (pair ? 2 : 1));
        id2s[rb] = id2;
    }
    // the uncertified rows go to their lists -- open rows from the front, pair rows from the back of the same arrays.  ONE
    // atomic per wave and list, issued here; the slots are only needed after the output phase, which hides the round trip
    // (four dependent atomics in a row cost 20k cycles per workgroup)
    const unsigned long long balo0 = __ballot(cls[0] == 1), balo1 = __ballot(cls[1] == 1);
    const unsigned long long balp0 = __ballot(cls[0] == 2), balp1 = __ballot(cls[1] == 2);
    const int n_open = (int)(__popcll(balo0) + __popcll(balo1)), n_pair = (int)(__popcll(balp0) + __popcll(balp1));
    int base_o = 0, base_p = 0;
    if (lane == 0) {
        if (n_open) base_o = atomicAdd(a.flag_count, n_open);
        if (n_pair) base_p = atomicAdd(a.flag_count + 1, n_pair);
    }

    VQ_PHASE(3);   // idx + list written
    // ---- outputs: whole rows per wave instruction (lane l moves elements 4 l .. 4 l + 3), RU rows in flight; x is re-read
    //      (coalesced) for the squared error and the residual.  Rows of the exact passes are skipped in the loss here. ----
    double ds = 0.0;
    if (a.q_out || a.resid_out || a.sqerr_partial) {
#ifndef VQS16_RU_F32
#define VQS16_RU_F32 8
#endif
#ifndef VQS16_RU_BF16
#define VQS16_RU_BF16 16
#endif
        constexpr int RU = XF32 ? VQS16_RU_F32 : VQS16_RU_BF16;
        const bool want_x = a.sqerr_partial != nullptr || a.resid_out != nullptr;
        const bool lane_on = lane * 4 < DT;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const bool counted = row_ok[rb] && !flagged[rb] && (!a.row_mask || a.row_mask[rows[rb]] != 0);
            const unsigned long long cmask = __ballot(counted && half == 0);
#pragma unroll
            for (int r0 = 0; r0 < 32; r0 += RU) {
                if (!XF32) {
                    uint2 g[RU], xv[RU];
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int c = __builtin_amdgcn_readlane(code[rb], r0 + u);
                        const int64_t rr = wrow0 + rb * 32 + r0 + u;
                        if (lane_on) {
                            g[u] = *(const uint2 *)(a.embed_bf16 + (size_t)c * DT + lane * 4);
                            if (want_x) xv[u] = *(const uint2 *)((const unsigned short *)a.x + (rr < a.N ? rr : a.N - 1) * a.ldx + lane * 4);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int64_t rr = wrow0 + rb * 32 + r0 + u;
                        if (rr < a.N && lane_on) {
                            if (a.q_out) *(uint2 *)((unsigned short *)a.q_out + rr * a.ldq + lane * 4) = g[u];
                            if (a.resid_out) *(uint2 *)((unsigned short *)a.resid_out + rr * a.ldr + lane * 4) = vq_bf16x4_sub(xv[u], g[u]);
                        }
                        if (a.sqerr_partial && lane_on && ((cmask >> (r0 + u)) & 1ull)) {
                            const float d0 = __uint_as_float(g[u].x << 16) - __uint_as_float(xv[u].x << 16);
                            const float d1 = __uint_as_float(g[u].x & 0xffff0000u) - __uint_as_float(xv[u].x & 0xffff0000u);
                            const float d2 = __uint_as_float(g[u].y << 16) - __uint_as_float(xv[u].y << 16);
                            const float d3 = __uint_as_float(g[u].y & 0xffff0000u) - __uint_as_float(xv[u].y & 0xffff0000u);
                            ds += (double)(((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3);
                        }
                    }
                } else {
                    f32x4 g[RU], xv[RU];
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int c = __builtin_amdgcn_readlane(code[rb], r0 + u);
                        const int64_t rr = wrow0 + rb * 32 + r0 + u;
                        if (lane_on) {
                            g[u] = *(const f32x4 *)(a.embed + (size_t)c * DT + lane * 4);
                            if (want_x) xv[u] = *(const f32x4 *)((const float *)a.x + (rr < a.N ? rr : a.N - 1) * a.ldx + lane * 4);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int64_t rr = wrow0 + rb * 32 + r0 + u;
                        if (rr < a.N && lane_on) {
                            if (a.q_out) *(f32x4 *)((float *)a.q_out + rr * a.ldq + lane * 4) = g[u];
                            if (a.resid_out) *(f32x4 *)((float *)a.resid_out + rr * a.ldr + lane * 4) = xv[u] - g[u];
                        }
                        if (a.sqerr_partial && lane_on && ((cmask >> (r0 + u)) & 1ull)) {
                            const float d0 = g[u].x - xv[u].x, d1 = g[u].y - xv[u].y, d2 = g[u].z - xv[u].z, d3 = g[u].w - xv[u].w;
                            ds += (double)(((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3);
                        }
                    }
                }
            }
        }
    }
    if (n_open) {
        const int bo = __builtin_amdgcn_readfirstlane(base_o);
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const unsigned long long bal = rb ? balo1 : balo0;
            if (cls[rb] == 1) {
                const int slot = bo + (rb ? (int)__popcll(balo0) : 0) + (int)__popcll(bal & below);
                a.flag_rows[slot] = (int)rows[rb];
                a.flag_keys[slot] = ~0ull;
            }
        }
    }
    if (n_pair) {
        const int bp = __builtin_amdgcn_readfirstlane(base_p);
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const unsigned long long bal = rb ? balp1 : balp0;
            if (cls[rb] == 2) {
                const int64_t slot = a.N - 1 - (bp + (rb ? (int)__popcll(balp0) : 0) + (int)__popcll(bal & below));
                a.flag_rows[slot] = (int)rows[rb];
                a.flag_keys[slot] = (unsigned long long)(unsigned)code[rb] | ((unsigned long long)(unsigned)id2s[rb] << 32);
            }
        }
    }
    VQ_PHASE(4);
    if (a.sqerr_partial) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ds += __shfl_xor(ds, o, 64);
        __syncthreads();
        double *red = (double *)smem;
        if (lane == 0) red[wave] = ds;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < VQS_WAVES; ++w) t += red[w];
            a.sqerr_partial[blockIdx.x] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 rows through the fp16 screen (vq_screen16_f32_kernel).  x' = x 2^SX is split into two fp16 parts,
// x' = x_h + x_m + r with |r| <= 2^-20 |x'| (two truncating conversions, the first remainder is exact in fp32), and both
// parts are multiplied with the ONE fp16 codebook part: 2 MFMAs per k-step instead of the 3 of the bf16 kernel below, one
// A-fragment read for the two.  One 32-row block per wave (x_h and x_m fill the registers the bf16-row kernel spends on a
// second row block), VQS_F32_WAVES waves per workgroup.  The skew is over tiles here: the top-3 epilogue of tile t - 1
// (accumulator `pa`) is issued in slices between the 2 NK MFMAs of tile t.  Bound: the bf16-row kernel's terms with
// 2 D + 1 accumulated terms, the dropped remainder 2 * 2^-20 X Y and two truncated operand sets in `conv`.
// q (fp32 code rows from `embed`), residual and squared error come from the row-cooperative pass that re-reads x.
// ------------------------------------------------------------------------------------------------
#ifndef VQS_F32_WAVES
#define VQS_F32_WAVES 8
#endif
#define VQ_SCREEN_F32_ROWS (VQS_F32_WAVES * 32)

template <int DT> struct Screen16F32Cfg {
    static constexpr int TILE_B = 64 * DT + 1024;
    static constexpr int SUB = Screen16Cfg<DT>::SUB;
    static constexpr int BUF_B = (SUB * TILE_B / 1024 + VQS_F32_WAVES - 1) / VQS_F32_WAVES * VQS_F32_WAVES * 1024;
    static constexpr int SMEM = 2 * BUF_B;
};

// One 32-row block per wave, NPART fp16 operand sets per row:
//   XBF16 = false, NPART = 2 : fp32 rows, D <= 256 (x_h + x_m, described above)
//   XBF16 = false, NPART = 1 : fp32 rows, D = 512 -- the registers hold ONE operand set, x_h = fp16_rne(x'); the certificate
//                              charges the measured residual, |(x' - x_h).c| <= ||x' - x_h|| Y per code (computed per row)
//   XBF16 = true,  NPART = 1 : bf16 rows, D = 512 -- exact operands like vq_screen16_kernel, half its rows per wave
template <int DT, int METRIC, bool XBF16, int NPART>
__global__ void __launch_bounds__(VQS_F32_WAVES * 64, (XBF16 && NPART == 1 && DT <= 256) ? 4 : 8 / VQS_F32_WAVES) vq_screen16_1rb_kernel(const ScreenArgs a0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ScreenArgs a = vq_head_screen_args(a0);
    using Cfg = Screen16F32Cfg<DT>;
    constexpr int W = VQS_F32_WAVES;
    constexpr int TILE_B = Cfg::TILE_B;
    constexpr int SUB = Cfg::SUB;
    constexpr int SUPER_B = SUB * TILE_B;
    constexpr int BUF_B = Cfg::BUF_B;
    constexpr int NCHUNK = SUPER_B / 1024;
    constexpr int NK = DT / 16;
    constexpr int TS = NPART * NK;                  // MFMAs (steps) per tile
    constexpr int PMAX = (NCHUNK + W - 1) / W;      // pieces per wave and buffer
    constexpr int PPS = (PMAX + SUB - 1) / SUB;     // pieces a wave copies during one tile
    constexpr int NBATCH = (TS >= 4) ? 2 : 1;       // staging batches per tile
    constexpr int BS2 = (PPS + NBATCH - 1) / NBATCH;
    constexpr int HALF = TS / NBATCH;               // steps between batch starts
    constexpr int LAG2 = (HALF >= 8) ? HALF - 2 : HALF - 1;
    constexpr int PSTRIDE = W * 1024;
    constexpr int ES = XBF16 ? 2 : 4;               // element size of x / q
    static_assert(VQ_F16_TILE_GROUP % SUB == 0, "tile padding must cover the tiles of one barrier");
    static_assert(!(XBF16 && NPART != 1), "bf16 rows are exact fp16 operands: one set");
    static_assert(TS >= 2 && LAG2 >= 1, "staging schedule");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int half = lane >> 5;
    const int64_t wrow0 = (int64_t)blockIdx.x * VQ_SCREEN_F32_ROWS + wave * 32;
    const int piece_off = wave * 1024 + lane * 16;
    const char *tiles = a.tiles16;

    // ---- x rows: lane (j, half) holds features 16 ks + 8 half + 0..7 of its row for every k-step ----
    const int64_t row = wrow0 + j;
    const bool row_ok = row < a.N;
    constexpr bool KEEP = !XBF16 && DT <= 256;      // fp32 rows of 512 features do not fit the registers raw (256 of them): two passes
    uint4 xh[NK], xm[NPART == 2 ? NK : 1];
    f32x4 xr[KEEP ? NK : 1][2];
    const float *xrow = (const float *)a.x + (row_ok ? row : (a.N - 1)) * a.ldx + 8 * half;
    if (XBF16) {
        const unsigned short *p = (const unsigned short *)a.x + (row_ok ? row : (a.N - 1)) * a.ldx + 8 * half;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) xh[ks] = *(const uint4 *)(p + ks * 16);
    } else if (KEEP) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) { xr[ks][0] = *(const f32x4 *)(xrow + ks * 16); xr[ks][1] = *(const f32x4 *)(xrow + ks * 16 + 4); }
    }
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
        *(f32x4 *)(smem + piece_off + k * PSTRIDE) = *(const f32x4 *)(tiles + piece_off + (size_t)k * PSTRIDE);

    float xs2;
    {
        float xs = 0.f;
        if (XBF16) {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const unsigned w[4] = {xh[ks].x, xh[ks].y, xh[ks].z, xh[ks].w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[q]), __builtin_bit_cast(bf16x2, w[q]), xs, false);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 v = KEEP ? xr[KEEP ? ks : 0][q] : *(const f32x4 *)(xrow + ks * 16 + 4 * q);
                    xs = __builtin_fmaf(v.x, v.x, xs); xs = __builtin_fmaf(v.y, v.y, xs);
                    xs = __builtin_fmaf(v.z, v.z, xs); xs = __builtin_fmaf(v.w, v.w, xs);
                    if (!KEEP && q == 1 && (ks & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
        }
        xs += __shfl_xor(xs, 32, 64);
        xs2 = xs * 1.001f;
    }
    // scale exponents: as in vq_screen16_kernel (largest finite row norm of the wave below 2^14)
    unsigned mx;
    {
        const unsigned bits = __float_as_uint(xs2);
        mx = (bits & 0x7f800000u) == 0x7f800000u ? 0u : (bits & 0x7fffffffu);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)mx, o, 64); mx = t > mx ? t : mx; }
    mx = (unsigned)__builtin_amdgcn_readfirstlane((int)mx);
    const int sc = (int)a.scalars[2];
    const int e2 = (int)(mx >> 23) - 127;
    int SX = (mx == 0u) ? 0 : 14 - ((e2 >> 1) + 1);
    SX = SX > 120 - sc ? 120 - sc : SX;
    SX = SX < -120 - sc ? -120 - sc : SX;
    SX = SX > sc + 90 ? sc + 90 : SX;
    SX = SX > 126 ? 126 : (SX < -126 ? -126 : SX);
    const float S = __uint_as_float((unsigned)(SX + 127) << 23);
    const float iS = __uint_as_float((unsigned)(127 - SX) << 23);
    const float SS = __uint_as_float((unsigned)(SX + sc + 127) << 23);
    const float iSS = __uint_as_float((unsigned)(127 - SX - sc) << 23);

    // ---- rows -> scaled fp16 operand set(s) ----
    float rx2 = 0.f;   // fp32 rows, one operand set: ||x' - x_h||^2 (scaled units)
    if (XBF16) {
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            unsigned w[4] = {xh[ks].x, xh[ks].y, xh[ks].z, xh[ks].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = __uint_as_float(w[q] << 16) * S, hi = __uint_as_float(w[q] & 0xffff0000u) * S;
                w[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));   // exact above 2^-14, truncated below (conv)
            }
            xh[ks] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const f32x4 w0 = KEEP ? xr[KEEP ? ks : 0][0] : *(const f32x4 *)(xrow + ks * 16);        // second pass over the row (L2)
            const f32x4 w1 = KEEP ? xr[KEEP ? ks : 0][1] : *(const f32x4 *)(xrow + ks * 16 + 4);
            const float v[8] = {w0.x * S, w0.y * S, w0.z * S, w0.w * S, w1.x * S, w1.y * S, w1.z * S, w1.w * S};
            unsigned hw[4], mw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (NPART == 2) {      // x_h: truncated x', x_m: truncated exact remainder
                    const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v[2 * q], v[2 * q + 1]));
                    const float r0 = v[2 * q] - (float)h[0], r1 = v[2 * q + 1] - (float)h[1];   // exact: the low bits of x'
                    hw[q] = __builtin_bit_cast(unsigned, h);
                    mw[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0, r1));
                } else {               // one set, rounded to nearest; the residual is measured, not modelled
                    f16x2 h;
                    h[0] = (_Float16)v[2 * q]; h[1] = (_Float16)v[2 * q + 1];
                    const float r0 = v[2 * q] - (float)h[0], r1 = v[2 * q + 1] - (float)h[1];
                    rx2 = __builtin_fmaf(r0, r0, rx2); rx2 = __builtin_fmaf(r1, r1, rx2);
                    hw[q] = __builtin_bit_cast(unsigned, h);
                    mw[q] = 0u;
                }
            }
            xh[ks] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
            if (NPART == 2) xm[ks] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
            if (!KEEP && (ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // 4 k-steps of raw loads in flight, not all 32 (256 registers)
        }
        if (NPART == 1) rx2 += __shfl_xor(rx2, 32, 64);
    }

    // the certificate's pieces, per code (vq_screen16_kernel above): e(c) = Rrow + Arow ||c|| + kb ||c||^2
    float Arow, Rrow, ArowS;
    {
        const float rho = __uint_as_float(a.scalars[1]);
        const float r0 = __uint_as_float(a.scalars[3]);
        const float u = 5.9604645e-8f;   // 2^-24
        const float conv = NPART * u * sqrtf((float)DT) * iS;                                    // truncated elements of each operand set
        const float xn = sqrtf(xs2) * 1.0001f;
        // the part of x' the operand sets do not carry: 2^-20 X ||c|| (two sets) / the measured ||x' - x_h|| ||c|| (one set)
        float drop = 0.f;
        if (!XBF16 && NPART == 2) drop = 9.5367432e-7f * 1.01f * xn;
        if (!XBF16 && NPART == 1) drop = sqrtf(rx2 * 1.001f) * 1.001f * iS;
        const float nacc = (float)(NPART * DT + 1);                                               // accumulated terms
        if (METRIC == 0) {
            Arow = (xn * (u * (10.f + (float)DT + 2.002f * nacc) + rho) + drop + conv) * (1.0001f + 4.f * nacc * u);      // (+ the MFMA's rounding of the added part)
            Rrow = 5.f * u * xs2 + xn * r0 + 2e-8f;
        } else {
            Arow = (xn * (u * (float)(DT + 2 * NPART * DT) * 1.001f + rho) + drop + conv) * (1.0001f + 4.f * nacc * u);
            Rrow = xn * r0 + 1e-30f;
        }
        ArowS = Arow * SS;
        const bool y2ok = __uint_as_float(a.scalars[0]) < 1e37f;       // overflow guard, as in vq_screen16_kernel
        const bool ok = (Arow < 1e30f) & (ArowS < 1e30f) & (Rrow < 1e30f) & y2ok;
        Arow = ok ? Arow : 0.f; ArowS = ok ? ArowS : 0.f; Rrow = ok ? Rrow : __builtin_inff();
    }

    float m1 = -__builtin_inff(), m2 = -__builtin_inff(), m3 = -__builtin_inff();
    int tix = 0, tix2 = 0;
    f32x16 pa;                                            // the previous tile's scores
#pragma unroll
    for (int r = 0; r < 16; ++r) pa[r] = -3.0e38f;       // "a padding code": never wins against a real one
    int ptile = 0;
    auto fold = [&](const f32x16 &acc, int e) {
        const float k = __uint_as_float((__float_as_uint(acc[e]) & 0xfffffff0u) | (unsigned)e);
        asm volatile("v_med3_f32 %2, %1, %2, %3\n\tv_med3_f32 %1, %0, %1, %3\n\tv_max_f32 %0, %0, %3"
                     : "+v"(m1), "+v"(m2), "+v"(m3) : "v"(k));
    };
    auto book = [&](float om1, float om2, int tile_id) {
        const bool c1 = m1 != om1;
        const int from_old_best = (c1 && m2 == om1) ? tix : tile_id;
        tix2 = (m2 != om2) ? from_old_best : tix2;
        tix = c1 ? tile_id : tix;
    };

    const int nst = a.n_tiles16 / SUB;
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        __syncthreads();
        const char *sbase = smem + buf * BUF_B;
        const bool more = st + 1 < nst;
        const char *gsrc = tiles + (size_t)(more ? st + 1 : st) * SUPER_B + piece_off;
        char *ldst = smem + (buf ^ 1) * BUF_B + piece_off;
#pragma unroll 1
        for (int sub = 0; sub < SUB; ++sub) {
            const char *tile = sbase + sub * TILE_B;
            const float *nh = (const float *)(tile + 64 * DT);
            const int tile_id = st * SUB + sub;
            const bool has_pad = (tile_id + 1) * 32 > a.C;
            const uint4 *ap = (const uint4 *)tile + lane;
            uint4 af[VQS16_PF];
            f32x4 stg[BS2];
            f32x16 acc;
            const int p0 = sub * PPS;
            const float o1 = m1, o2 = m2;
#pragma unroll
            for (int p = 0; p < VQS16_PF; ++p) af[p] = ap[(p % NK) * 64];
#pragma unroll
            for (int s = 0; s < TS; ++s) {
                const int ks = s / NPART;
                const f16x8 av = __builtin_bit_cast(f16x8, af[ks % VQS16_PF]);
                const f16x8 bv = __builtin_bit_cast(f16x8, (NPART == 2 && (s & 1)) ? xm[NPART == 2 ? ks : 0] : xh[ks]);
                if (s == 0) {
                    f32x16 init;        // start value + (row factor) x ||c||: an upper bound of the code's true score
                    {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = *(const f32x4 *)(nh + 8 * q + 4 * half);
                            const f32x4 w = *(const f32x4 *)(nh + 32 + 8 * q + 4 * half);
                            if (METRIC != 0) { v.x = v.x < -1e38f ? v.x : 0.f; v.y = v.y < -1e38f ? v.y : 0.f;
                                               v.z = v.z < -1e38f ? v.z : 0.f; v.w = v.w < -1e38f ? v.w : 0.f; }
                            const f32x2 vp[2] = {f32x2{v.x, v.y}, f32x2{v.z, v.w}}, wp[2] = {f32x2{w.x, w.y}, f32x2{w.z, w.w}};
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                const f32x2 r = (METRIC == 0 || has_pad) ? __builtin_elementwise_fma(wp[i], f32x2{ArowS, ArowS}, vp[i] * f32x2{SS, SS})
                                                                         : wp[i] * f32x2{ArowS, ArowS};
                                init[4 * q + 2 * i] = r.x; init[4 * q + 2 * i + 1] = r.y;
                            }
                        }
                        if (has_pad) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) init[r] = fmaxf(init[r], -3.0e38f);
                        }
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, init, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
                }
                if ((s % NPART) == NPART - 1 && ks + VQS16_PF < NK) af[ks % VQS16_PF] = ap[(ks + VQS16_PF) * 64];
                if (s >= 1) {   // epilogue slice of the previous tile: 16 scores over the steps 1 .. TS - 1
#pragma unroll
                    for (int e = (s - 1) * 16 / (TS - 1); e < s * 16 / (TS - 1); ++e) fold(pa, e);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int bt = 0; bt < NBATCH; ++bt) {
                    if (s == bt * HALF) {
#pragma unroll
                        for (int i = 0; i < BS2; ++i)
                            if (bt * BS2 + i < PPS) {
                                const int pc = min(p0 + bt * BS2 + i, PMAX - 1);
                                stg[i] = *(const f32x4 *)(gsrc + (size_t)pc * PSTRIDE);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (s == bt * HALF + LAG2) {
#pragma unroll
                        for (int i = 0; i < BS2; ++i)
                            if (bt * BS2 + i < PPS) *(f32x4 *)(ldst + min(p0 + bt * BS2 + i, PMAX - 1) * PSTRIDE) = stg[i];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            book(o1, o2, ptile);
            pa = acc;
            ptile = tile_id;
        }
    }
    {
        const float o1 = m1, o2 = m2;
#pragma unroll
        for (int e = 0; e < 16; ++e) fold(pa, e);
        book(o1, o2, ptile);
    }

    // ---- merge the half-waves, classify (certified / pair / open), emit ----
    int code, cls, id2;
    bool flagged;
    {
        const int e1 = (int)(__float_as_uint(m1) & 15u), e2b = (int)(__float_as_uint(m2) & 15u);
        const int ia1 = tix * 32 + 8 * (e1 >> 2) + 4 * half + (e1 & 3);
        const int ia2 = tix2 * 32 + 8 * (e2b >> 2) + 4 * half + (e2b & 3);
        const float a1 = m1, a2 = m2, a3 = m3;
        const float p1 = __shfl_xor(a1, 32, 64), p2 = __shfl_xor(a2, 32, 64), p3 = __shfl_xor(a3, 32, 64);
        const int ib1 = __shfl_xor(ia1, 32, 64), ib2 = __shfl_xor(ia2, 32, 64);
        const bool take = p1 > a1;
        const float h1 = take ? p1 : a1, h2 = take ? p2 : a2, h3 = take ? p3 : a3;
        const float l1 = take ? a1 : p1, l2 = take ? a2 : p2;
        const int ih1 = take ? ib1 : ia1, ih2 = take ? ib2 : ia2, il1 = take ? ia1 : ib1;
        const float b1 = h1;
        const bool second_low = l1 > h2;
        const float b2 = second_low ? l1 : h2;
        id2 = second_low ? il1 : ih2;
        const float b3 = second_low ? fmaxf(h2, l2) : fmaxf(h3, l1);
        code = ih1;
        const float kb = 5.9604645e-8f * (5.f + 1.001f * (float)(DT + 1) + 0.51f) * 1.001f;
        auto code_part = [&](int c) {       // A ||c|| + kb ||c||^2 of one code (||c|| from the tile tail of the packed codebook)
            const int cc = c < a.C ? c : 0;
            const float yb = *(const float *)(a.tiles16 + (size_t)(cc >> 5) * TILE_B + 64 * DT + (32 + (cc & 31)) * 4);
            return (Arow * yb + (METRIC == 0 ? kb * yb * yb * 1.001f : 0.f)) * 1.0001f;
        };
        const float cp1 = code_part(ih1);
        const float thr = (2.f * cp1 + 2.f * Rrow) * SS + 8e-6f * fabsf(b1);
        const bool certified = ((b1 - b2) > thr) && code < a.C && b1 < 3.0e38f;
        const bool pair = !certified && ((b1 - b3) > thr) && code < a.C && id2 < a.C && b1 < 3.0e38f;
        flagged = !certified;
        if (row_ok && half == 0) {
            a.idx_out[row * a.idx_stride] = (int64_t)(code < a.C ? code : 0);
            if (a.dbg) {        // (in the units of t = x.c - ||c||^2 / 2, what the sweep added taken off again: certified <=> d[0] - d[1] > d[2])
                float *d = a.dbg + row * 4;
                float add[2], sc2[2];
                const int cs[2] = {ih1, id2};
                const float us[2] = {b1, b2};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int cc = cs[k] < a.C ? cs[k] : 0;
                    const float *tl = (const float *)(a.tiles16 + (size_t)(cc >> 5) * TILE_B + 64 * DT) + (cc & 31);
                    const float v = METRIC == 0 ? tl[0] : 0.f, w = tl[32], pz = METRIC == 0 ? tl[64] : 0.f;
                    const float init = METRIC == 0 ? __builtin_fmaf(w, ArowS, v * SS) : w * ArowS;
                    sc2[k] = (us[k] - init) * iSS + pz;
                    add[k] = init * iSS - pz;
                }
                d[0] = sc2[0]; d[1] = sc2[1]; d[2] = thr * iSS - add[0] + add[1]; d[3] = certified ? 0.f : (pair ? 2.f : 1.f);
            }
        }
        if (code >= a.C) code = 0;
        const bool on = row_ok && half == 0;
        cls = !on ? 0 : (certified ? 0 : (pair ? 2 : 1));
    }
    const unsigned long long balo = __ballot(cls == 1), balp = __ballot(cls == 2);
    int base_o = 0, base_p = 0;
    if (lane == 0) {
        if (balo) base_o = atomicAdd(a.flag_count, (int)__popcll(balo));
        if (balp) base_p = atomicAdd(a.flag_count + 1, (int)__popcll(balp));
    }

    // ---- q rows, residual and squared error: whole rows per wave instruction (lane l moves elements c0 + 4 l .. + 3 of a
    //      256-element chunk), 8 rows in flight; x is re-read (coalesced).  Rows of the exact passes are skipped in the loss. ----
    if (a.q_out || a.sqerr_partial || a.resid_out) {
        const int counted = (row_ok && !flagged && (!a.row_mask || a.row_mask[row] != 0)) ? 1 : 0;
        const unsigned long long cmask = __ballot(counted && half == 0);
        const bool want_sq = a.sqerr_partial != nullptr;
        const bool want_x = want_sq || a.resid_out != nullptr;
        double ds = 0.0;
        const char *codes = XBF16 ? (const char *)a.embed_bf16 : (const char *)a.embed;
#pragma unroll
        for (int c0 = 0; c0 < DT; c0 += 256) {
            const bool lane_on = c0 + lane * 4 < DT;
#pragma unroll
            for (int r0 = 0; r0 < 32; r0 += 8) {
                f32x4 g[8], xv[8];     // bf16: the low two dwords carry the 4 elements
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = __builtin_amdgcn_readlane(code, r0 + u);
                    const int64_t rr = wrow0 + r0 + u;
                    if (lane_on) {
                        const char *gp = codes + ((size_t)c * DT + c0 + lane * 4) * ES;
                        const char *xp = (const char *)a.x + ((rr < a.N ? rr : a.N - 1) * a.ldx + c0 + lane * 4) * ES;
                        if (XBF16) {
                            const uint2 t = *(const uint2 *)gp; g[u].x = __uint_as_float(t.x); g[u].y = __uint_as_float(t.y);
                            if (want_x) { const uint2 t2 = *(const uint2 *)xp; xv[u].x = __uint_as_float(t2.x); xv[u].y = __uint_as_float(t2.y); }
                        } else {
                            g[u] = *(const f32x4 *)gp;
                            if (want_x) xv[u] = *(const f32x4 *)xp;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t rr = wrow0 + r0 + u;
                    if (lane_on) {
                        float d0, d1, d2, d3;
                        if (XBF16) {
                            const uint2 gb = make_uint2(__float_as_uint(g[u].x), __float_as_uint(g[u].y));
                            const uint2 xb2 = make_uint2(__float_as_uint(xv[u].x), __float_as_uint(xv[u].y));
                            if (a.q_out && rr < a.N) *(uint2 *)((unsigned short *)a.q_out + rr * a.ldq + c0 + lane * 4) = gb;
                            if (a.resid_out && rr < a.N) *(uint2 *)((unsigned short *)a.resid_out + rr * a.ldr + c0 + lane * 4) = vq_bf16x4_sub(xb2, gb);
                            d0 = __uint_as_float(gb.x << 16) - __uint_as_float(xb2.x << 16);
                            d1 = __uint_as_float(gb.x & 0xffff0000u) - __uint_as_float(xb2.x & 0xffff0000u);
                            d2 = __uint_as_float(gb.y << 16) - __uint_as_float(xb2.y << 16);
                            d3 = __uint_as_float(gb.y & 0xffff0000u) - __uint_as_float(xb2.y & 0xffff0000u);
                        } else {
                            if (a.q_out && rr < a.N) *(f32x4 *)((float *)a.q_out + rr * a.ldq + c0 + lane * 4) = g[u];
                            if (a.resid_out && rr < a.N) *(f32x4 *)((float *)a.resid_out + rr * a.ldr + c0 + lane * 4) = xv[u] - g[u];
                            d0 = g[u].x - xv[u].x; d1 = g[u].y - xv[u].y; d2 = g[u].z - xv[u].z; d3 = g[u].w - xv[u].w;
                        }
                        if (want_sq && ((cmask >> (r0 + u)) & 1ull)) ds += (double)(((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3);
                    }
                }
            }
        }
        if (a.sqerr_partial) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ds += __shfl_xor(ds, o, 64);
            __syncthreads();
            double *red = (double *)smem;
            if (lane == 0) red[wave] = ds;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < W; ++w) t += red[w];
                a.sqerr_partial[blockIdx.x] = t;
            }
        }
    }
    const int bo = __builtin_amdgcn_readfirstlane(base_o), bp = __builtin_amdgcn_readfirstlane(base_p);   // lane 0's, read with all lanes active
    if (balo && cls == 1) {
        const int slot = bo + (int)__popcll(balo & ((1ull << lane) - 1ull));
        a.flag_rows[slot] = (int)row;
        a.flag_keys[slot] = ~0ull;
    }
    if (balp && cls == 2) {
        const int64_t slot = a.N - 1 - (bp + (int)__popcll(balp & ((1ull << lane) - 1ull)));
        a.flag_rows[slot] = (int)row;
        a.flag_keys[slot] = (unsigned long long)(unsigned)code | ((unsigned long long)(unsigned)id2 << 32);
    }
}

static inline int64_t screen_rows_per_block(int x_dtype) { return x_dtype == VQHIP_BF16 ? VQ_SCREEN_ROWS : VQ_SCREEN_F32_ROWS; }

extern "C" int64_t vqhip_screen_blocks(int64_t N, int x_dtype)
{
    const int64_t r = screen_rows_per_block(x_dtype);
    return N <= 0 ? 0 : (N + r - 1) / r;
}

extern "C" int64_t vqhip_screen_partials(int64_t N, int x_dtype)
{
    return N <= 0 ? 0 : vqhip_screen_blocks(N, x_dtype) + VQ_FINISH_BLOCKS;
}

extern "C" size_t vqhip_screen_workspace_bytes(int64_t N)
{
    // 16-byte header (count) | ceil(N / 128) ints (round 6: arrival counters of the exact sweep's 128-row chunks, vq_tail_kernel; zeroed
    // with the header by the residual chain, unused otherwise) | N ints (row list, padded to 8 bytes) | N u64 (keys of the exact pass)
    // | segmented staging of the persistent screening kernel: 2 VQ_SEG_MAX counters, (N + 256 VQ_SEG_MAX) u64 keys and as many int rows
    if (N <= 0) return 0;
    const size_t nseg = (size_t)N + 256 * (size_t)VQ_SEG_MAX;
    return 16 + vq_screen_done_ints(N) * sizeof(int) + (((size_t)N * sizeof(int) + 7) & ~(size_t)7) + (size_t)N * sizeof(unsigned long long)
         + 2 * VQ_SEG_MAX * sizeof(int) + nseg * sizeof(unsigned long long) + nseg * sizeof(int);
}

extern "C" int vqhip_screen_supported(int64_t N, int D, int C)
{
    return (D == 32 || D == 64 || D == 128 || D == 256 || D == 512) && N > 0 && N < ((int64_t)1 << 31) - 512 && C >= 2;
}

template <int DT, int METRIC>
static int launch_screen(const ScreenArgs &a, int x_dtype, hipStream_t st)
{
    const unsigned blocks = (unsigned)vqhip_screen_blocks(a.N, x_dtype);
    // fp16 single-codebook-part kernels: bf16 rows with D <= 256 keep two row blocks per wave, everything else one
    if (x_dtype == VQHIP_BF16) {
        if constexpr (DT == 256) {   // persistent form with the cyclic tile stream (vq_screen_c.hip): large batches without squared-error / residual outputs
            if (vq_screenc_eligible(a, x_dtype, DT)) return vq_screenc_launch(a, METRIC, st);
        }
        if constexpr (DT <= 256) {
            static VqAttrOnce once;
#ifndef VQS16_LDS_PAD
#define VQS16_LDS_PAD 0      // A/B: extra dynamic LDS (> 8 KiB at D = 256: one workgroup per CU, i.e. one wave per SIMD)
#endif
            constexpr int SMEM16 = Screen16Cfg<DT>::SMEM + VQS16_LDS_PAD;
            if (int rc = vq_set_max_smem(once, (const void *)vq_screen16_kernel<DT, METRIC>, SMEM16, "vq_screen16_kernel")) return rc;
            hipLaunchKernelGGL((vq_screen16_kernel<DT, METRIC>), dim3(blocks, (unsigned)(a.heads > 1 ? a.heads : 1)), dim3(VQS_WAVES * 64), SMEM16, st, a);
        } else {
            static VqAttrOnce once;
            constexpr int SMEM16 = Screen16F32Cfg<DT>::SMEM;
            if (int rc = vq_set_max_smem(once, (const void *)vq_screen16_1rb_kernel<DT, METRIC, true, 1>, SMEM16, "vq_screen16_1rb_kernel")) return rc;
            hipLaunchKernelGGL((vq_screen16_1rb_kernel<DT, METRIC, true, 1>), dim3(blocks, (unsigned)(a.heads > 1 ? a.heads : 1)), dim3(VQS_F32_WAVES * 64), SMEM16, st, a);
        }
    } else {
        // fp32 rows, D <= 256: one fp16 operand set, two row blocks per wave.  (The two-operand-set form x_h + x_m of the one-row-block
        // kernel -- NPART = 2, half the uncertified rows -- was measured slower in rounds 2 and 3, 1.43 vs 1.21 ms at cfg-2 size, and is
        // no longer instantiated.)
        if constexpr (DT <= 256) {
            {
                static VqAttrOnce once;
                constexpr int SMEM16 = Screen16Cfg<DT>::SMEM;
                // (NPART = 2 fits the registers for D <= 128 and halves the uncertified rows of fp32 inputs, but its second MFMA per
                //  k-step costs more than the exact passes it saves: cfg 5 26.0 vs 23.5 ms -- measured, not adopted)
                constexpr int NP = 1;
                if (int rc = vq_set_max_smem(once, (const void *)vq_screen16_kernel<DT, METRIC, true, NP>, SMEM16, "vq_screen16_kernel (fp32 rows)")) return rc;
                hipLaunchKernelGGL((vq_screen16_kernel<DT, METRIC, true, NP>), dim3(blocks, (unsigned)(a.heads > 1 ? a.heads : 1)), dim3(VQS_WAVES * 64), SMEM16, st, a);
                return vq_launch_status("vq_screen16_kernel (fp32 rows)");
            }
        }
        if constexpr (DT > 256) {
            static VqAttrOnce once;
            constexpr int SMEM16 = Screen16F32Cfg<DT>::SMEM;
            if (int rc = vq_set_max_smem(once, (const void *)vq_screen16_1rb_kernel<DT, METRIC, false, 1>, SMEM16, "vq_screen16_1rb_kernel")) return rc;
            hipLaunchKernelGGL((vq_screen16_1rb_kernel<DT, METRIC, false, 1>), dim3(blocks, (unsigned)(a.heads > 1 ? a.heads : 1)), dim3(VQS_F32_WAVES * 64), SMEM16, st, a);
        }
    }
    return vq_launch_status("vq_screen16 kernels");
}

template <int DT>
static int dispatch_screen(const ScreenArgs &a, int x_dtype, int metric, hipStream_t st)
{
    return metric == VQHIP_EUCLID ? launch_screen<DT, 0>(a, x_dtype, st) : launch_screen<DT, 1>(a, x_dtype, st);
}


extern "C" int vqhip_assign_screened(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed,
                                     const float *embed, int C, int metric, int64_t *idx_out, void *q_out, int64_t ldq,
                                     void *resid_out, int64_t ldr, double *sqerr_partial, const uint8_t *row_mask,
                                     void *workspace, size_t workspace_bytes, float *debug_out, void *stream)
{
    return vq_assign_screened_impl(x, x_dtype, N, D, ldx, packed, embed, C, metric, idx_out, q_out, ldq, resid_out, ldr, sqerr_partial,
                                   row_mask, workspace, workspace_bytes, debug_out, nullptr, 0, stream);
}

// H searches in one set of launches (blockIdx.y = head): the heads of a multi-head VectorQuantize with separate codebooks
// (vqp.py:1044-1049), RandomProjectionQuantizer's 16 heads (random_projection_quantizer.py:37-59).  Every buffer of head h sits
// h strides behind head 0's: x at x_hstride ELEMENTS, packed at vqhip_packed_bytes(C, D), embed at C * D floats, idx_out at N, q_out at
// q_hstride elements, the workspace at vqhip_screen_batched_ws_stride(N) bytes.
extern "C" size_t vqhip_screen_batched_ws_stride(int64_t N)
{
    return (vqhip_screen_workspace_bytes(N) + 255) & ~(size_t)255;
}

extern "C" int vqhip_assign_screened_batched(const void *x, int x_dtype, int H, int64_t N, int D, int64_t ldx, int64_t x_hstride,
                                             const float *packed, const float *embed, int C, int metric, int64_t *idx_out,
                                             void *q_out, int64_t ldq, int64_t q_hstride, const uint8_t *row_mask,
                                             void *workspace, size_t workspace_bytes, void *stream)
{
    if (H < 1) VQ_FAIL(VQHIP_EINVAL, "assign_screened_batched: H < 1");
    const size_t wss = vqhip_screen_batched_ws_stride(N);
    if (workspace_bytes < wss * (size_t)H) VQ_FAIL(VQHIP_EINVAL, "assign_screened_batched: workspace too small");
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    if ((x_hstride * es) & 15) VQ_FAIL(VQHIP_EALIGN, "assign_screened_batched: heads' rows must stay 16-byte aligned");
    if (q_out && ((q_hstride * es) % (4 * es))) VQ_FAIL(VQHIP_EALIGN, "assign_screened_batched: heads' q rows must stay aligned to 4 elements");
    VqHeadStrides hs;
    hs.heads = H;
    hs.x = x_hstride * es;
    hs.packed = (int64_t)vqhip_packed_bytes(C, D);
    hs.embed = (int64_t)C * D * 4;
    hs.codes = (x_dtype == VQHIP_BF16) ? hs.packed : hs.embed;
    hs.idx = N * 8;
    hs.q = q_hstride * es;
    hs.ws = (int64_t)wss;
    return vq_assign_screened_impl(x, x_dtype, N, D, ldx, packed, embed, C, metric, idx_out, q_out, ldq, nullptr, D, nullptr, row_mask,
                                   workspace, wss, nullptr, nullptr, 0, stream, &hs);
}

extern "C" int vqhip_screen_chain_supported(int x_dtype, int D)
{
#ifdef VQS16_F32_DIRECT      // (A/B build whose fp32-row prologue has no chain step)
    return 0;
#endif
    return (x_dtype == VQHIP_F32 && (D == 32 || D == 64 || D == 128 || D == 256)) ? 1 : 0;
}

extern "C" int vqhip_assign_screened_chain(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed,
                                           const float *embed, int C, int metric, int64_t *idx_out, const uint8_t *row_mask,
                                           void *workspace, size_t workspace_bytes, const vqhip_chain_t *chain, void *stream)
{
    if (!chain) VQ_FAIL(VQHIP_EINVAL, "assign_screened_chain: chain is null");
    // (without a previous stage to subtract -- prev_idx null -- this is the plain screened search with the chain's index stride: any
    //  rows the screen takes, e.g. the bf16 / D = 512 stages of a residual loop whose inputs vqhip_route_residual wrote)
    if (chain->prev_idx && !vqhip_screen_chain_supported(x_dtype, D))
        VQ_FAIL(VQHIP_EINVAL, "assign_screened_chain: a chained stage takes fp32 rows, D in {32, 64, 128, 256} only");
    if (metric != VQHIP_EUCLID) VQ_FAIL(VQHIP_EINVAL, "assign_screened_chain: Euclidean metric only");
    if (chain->idx_stride < 1) VQ_FAIL(VQHIP_EINVAL, "assign_screened_chain: idx_stride < 1");
    if (chain->prev_idx) {
        if (!chain->prev_embed || !chain->x_out || chain->prev_idx_stride < 1 || chain->ldxo < D)
            VQ_FAIL(VQHIP_EINVAL, "assign_screened_chain: prev_idx needs prev_embed, x_out and valid strides");
        if ((((uintptr_t)chain->prev_embed) & 15) || (((uintptr_t)chain->x_out) & 15) || ((chain->ldxo * 4) & 15))
            VQ_FAIL(VQHIP_EALIGN, "assign_screened_chain: prev_embed / x_out rows must be 16-byte aligned");
        if (chain->route_mode != 0)
            VQ_FAIL(VQHIP_EINVAL, "assign_screened_chain: route_mode is gone (round 4): form a routed residual with vqhip_route_residual and pass it as x with prev_idx = NULL");
    }
    return vq_assign_screened_impl(x, x_dtype, N, D, ldx, packed, embed, C, metric, idx_out, nullptr, D, nullptr, D, nullptr, row_mask,
                                   workspace, workspace_bytes, nullptr, chain, chain->header_zeroed != 0, stream);
}

// header_zeroed: the caller has zeroed the first 16 bytes of the workspace (the list counters) on this stream already.
int vq_assign_screened_impl(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed,
                            const float *embed, int C, int metric, int64_t *idx_out, void *q_out, int64_t ldq,
                            void *resid_out, int64_t ldr, double *sqerr_partial, const uint8_t *row_mask,
                            void *workspace, size_t workspace_bytes, float *debug_out, const vqhip_chain_t *chain,
                            int header_zeroed, void *stream, const VqHeadStrides *hs)
{
    const int heads = (hs && hs->heads > 1) ? hs->heads : 1;
    if (heads > 1 && (resid_out || sqerr_partial || debug_out))
        VQ_FAIL(VQHIP_EINVAL, "assign_screened: a batched launch has index and q outputs only (and the residual chain's x_out)");
    if (N < 0 || C <= 0) VQ_FAIL(VQHIP_EINVAL, "assign_screened: N < 0 or C <= 0");
    if (N == 0) return 0;
    if (!x || !packed || !embed || !idx_out || !workspace) VQ_FAIL(VQHIP_EINVAL, "assign_screened: null pointer");
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "assign_screened: unknown x dtype %d", x_dtype);
    if (metric != VQHIP_EUCLID && metric != VQHIP_COSINE_PRENORM)
        VQ_FAIL(VQHIP_EINVAL, "assign_screened: metric %d (VQHIP_EUCLID, or VQHIP_COSINE_PRENORM on rows normalised by vqhip_l2norm_rows)", metric);
    if (!vqhip_screen_supported(N, D, C)) VQ_FAIL(VQHIP_EDIM, "assign_screened: N=%lld D=%d C=%d outside the screened path (D in {32,64,128,256,512}, C >= 2)", (long long)N, D, C);
    if (workspace_bytes < vqhip_screen_workspace_bytes(N)) VQ_FAIL(VQHIP_EINVAL, "assign_screened: workspace too small");
    if (ldx < D || (q_out && ldq < D) || (resid_out && ldr < D)) VQ_FAIL(VQHIP_EINVAL, "assign_screened: row stride smaller than D");
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    if ((((uintptr_t)packed) & 15) || (((uintptr_t)embed) & 15) || (((uintptr_t)workspace) & 7))
        VQ_FAIL(VQHIP_EALIGN, "assign_screened: packed / embed must be 16-byte aligned, workspace 8-byte aligned");
    if ((((uintptr_t)x) & 15) || ((ldx * es) & 15)) VQ_FAIL(VQHIP_EALIGN, "assign_screened: x rows must be 16-byte aligned");
    if (q_out && ((((uintptr_t)q_out) % (4 * es)) || ((ldq * es) % (4 * es)))) VQ_FAIL(VQHIP_EALIGN, "assign_screened: q rows must be aligned to 4 elements");
    if (resid_out && ((((uintptr_t)resid_out) % (4 * es)) || ((ldr * es) % (4 * es)))) VQ_FAIL(VQHIP_EALIGN, "assign_screened: residual rows must be aligned to 4 elements");

    hipStream_t st = (hipStream_t)stream;
    int *count = (int *)workspace;
    int *done = count + 4;                                           // arrival counters of the merged exact-pass launch (chain)
    int *rows = done + vq_screen_done_ints(N);
    if (!header_zeroed) {
        hipError_t e = heads > 1 ? hipMemset2DAsync(count, (size_t)hs->ws, 0, 16, (size_t)heads, st) : hipMemsetAsync(count, 0, 16, st);
        if (e != hipSuccess) VQ_FAIL((int)e, "assign_screened: hipMemsetAsync: %s", hipGetErrorString(e));
    }

    const char *base = (const char *)packed;
    ScreenArgs a;
    a.x = x; a.N = N; a.ldx = ldx;
    a.tiles16 = base + vq_packed_f16_offset(C, D);
    a.n_tiles16 = (int)vq_tiles16(C);
    a.embed_bf16 = (const unsigned short *)(base + vq_packed_bf16_offset(C, D));
    a.embed = embed;
    a.scalars = (const unsigned *)(base + vq_packed_scalars_offset(C, D));
    a.C = C;
    a.idx_out = idx_out; a.q_out = q_out; a.ldq = ldq; a.resid_out = resid_out; a.ldr = ldr;
    a.sqerr_partial = sqerr_partial; a.row_mask = row_mask;
    unsigned long long *keys = (unsigned long long *)((char *)rows + (((size_t)N * sizeof(int) + 7) & ~(size_t)7));
    a.flag_count = count; a.flag_rows = rows; a.flag_keys = keys; a.dbg = debug_out;
    {
        const size_t nseg = (size_t)N + 256 * (size_t)VQ_SEG_MAX;
        a.seg_counts = (int *)(keys + N);
        a.seg_keys = (unsigned long long *)(a.seg_counts + 2 * VQ_SEG_MAX);
        a.seg_rows = (int *)(a.seg_keys + nseg);
        a.seg_cap = 0;
    }
    a.heads = heads;
    a.hs_x = hs ? hs->x : 0; a.hs_packed = hs ? hs->packed : 0; a.hs_embed = hs ? hs->embed : 0;
    a.hs_idx = hs ? hs->idx : 0; a.hs_q = hs ? hs->q : 0; a.hs_ws = hs ? hs->ws : 0; a.hs_xo = hs ? hs->xo : 0;
    a.idx_stride = chain ? chain->idx_stride : 1;
    a.prev_idx = chain ? chain->prev_idx : nullptr;
    a.prev_idx_stride = chain ? chain->prev_idx_stride : 1;
    a.prev_embed = chain ? chain->prev_embed : nullptr;
    a.x_out = chain ? (float *)chain->x_out : nullptr;
    a.ldxo = chain ? chain->ldxo : 0;
    a.stagger = 0;
    a.stagger_first = 512;
#ifdef VQ_DEV_SWITCHES
    {   // dev builds only (make HIPFLAGS+=-DVQ_DEV_SWITCHES; A/B measurements of round 5, tools/time_chain_stage.py):
        // VQHIP_SCREEN_STAGGER=<k> start offset of every CU's second workgroup, x 1024 cycles;
        // VQHIP_CHAIN_NOWRITE=1 a chained stage does not store its input (TIMING ONLY: the exact passes then read stale rows)
        static int stag = -1, nowrite = -1;
        if (stag < 0) { const char *e = getenv("VQHIP_SCREEN_STAGGER"); stag = e ? atoi(e) : 0; }
        if (nowrite < 0) { const char *e = getenv("VQHIP_CHAIN_NOWRITE"); nowrite = (e && e[0] == '1') ? 1 : 0; }
        a.stagger = stag;
        if (nowrite) a.x_out = nullptr;
    }
#endif
#ifdef VQ_TRACE
    a.trace = vq_g_trace;
#endif
    int rc;
    switch (D) {
        case 32: rc = dispatch_screen<32>(a, x_dtype, metric, st); break;
        case 64: rc = dispatch_screen<64>(a, x_dtype, metric, st); break;
        case 128: rc = dispatch_screen<128>(a, x_dtype, metric, st); break;
        case 256: rc = dispatch_screen<256>(a, x_dtype, metric, st); break;
        default: rc = dispatch_screen<512>(a, x_dtype, metric, st); break;
    }
    if (rc) return rc;
#ifdef VQ_DEV_SWITCHES
    {   // VQHIP_SCREEN_ONLY=1 (dev tools that time the screening kernel by itself; dev builds only): leave the listed rows undecided
        static int only = -1;
        if (only < 0) { const char *e = getenv("VQHIP_SCREEN_ONLY"); only = (e && e[0] == '1') ? 1 : 0; }
        if (only) return 0;
    }
#endif
    const int with_pairs = 1;                                       // the screening kernels build the pair list as well
    // a chained stage's rows were materialised by its screening kernel: the exact passes read them there
    const bool chained = chain && chain->prev_idx;
    const void *xl = chained ? (const void *)chain->x_out : x;
    const int64_t ldl = chained ? chain->ldxo : ldx;
    VqHeadStrides hl;
    if (hs) { hl = *hs; if (chained) hl.x = hs->xo; }
    // the residual chain (index output only, counters zeroed with the header by its caller): ONE launch for the exact sweep of the open
    // rows and the two exact distances of the pair rows, both writing their index themselves (round 6; was refine -> pair -> finish)
    if (chain && header_zeroed && !q_out && !resid_out && !sqerr_partial && vq_tail_enabled())
        return vq_assign_listed_direct(xl, x_dtype, metric, N, D, ldl, packed, embed, C, idx_out, a.idx_stride, rows, count, keys, done, st,
                                       hs ? &hl : nullptr);
    return vq_assign_listed(xl, x_dtype, metric, N, D, ldl, packed, embed, C, idx_out, a.idx_stride, q_out, ldq, resid_out, ldr,
                            sqerr_partial ? sqerr_partial + vqhip_screen_blocks(N, x_dtype) : nullptr, row_mask, rows, count, keys,
                            with_pairs, st, hs ? &hl : nullptr);
}
