// vq_screen_c.hip -- persistent form of the single-pass fp16 screen (vq_screen16_kernel in vq_screen.hip) with a CYCLIC tile stream,
// gfx950 only.  Same certificate, same sweep (two 32-row blocks per wave, every A fragment from LDS feeds two MFMAs on different
// accumulators, the previous tile's top-3 fold between them), same outputs (index, q rows, open / pair lists).  What changes:
//
//   * TWO workgroups of 4 waves per CU (one wave of each per SIMD, 256 registers each) that stay for the whole launch.  Each streams
//     the codebook tiles through its own LDS buffers as an endless cycle (interval g of a workgroup holds tiles 2 (g mod nst), +1;
//     LDS-DMA, two buffers), paced by one barrier per interval.
//   * A workgroup processes its 256-row blocks as   nst sweep intervals  ->  3 "break" intervals (no MFMAs: classify + index + list
//     slots, q rows, list entries, request the next rows | wait for them | norms, scale, conversion)  -> ...   A sweep may start at ANY
//     interval of the cycle -- it just has to see nst consecutive intervals.  The second workgroup of every CU starts half a
//     period late and both have the same period, so one workgroup's break faces the other's sweep, never its break; the CUs start
//     spread over a fraction of a period, so that the chip's row traffic (the HBM-bound part of vq_screen16_kernel, where all
//     workgroups load, sweep and write in the same phase of every round) is spread over the whole launch.
//
// Eligibility (launch_screen in vq_screen.hip): bf16 rows, D = 256, no residual / squared-error output, N >= VQC_MIN_ROWS -- the
// default there since round 4 (full adversarial fuzz of tests/test_gpu_screen_fuzz.py green; VQHIP_SCREEN_PERSIST=0: the 4-wave kernel).
// Reference arithmetic that the screen certifies: cdist at vqp.py:58-62, argmax at vqp.py:140 (cosine: einsum at vqp.py:741).

#include <type_traits>
#include <utility>

#include "vq_screen_args.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DT> struct ScreenCCfg {
    static constexpr int W = 4;                        // waves per workgroup
    static constexpr int NK = DT / 16;                 // MFMA k-steps per tile
    static constexpr int TILE_B = 64 * DT + 1024;      // fp16 tile: 32 codes x DT + 32 floats -||c||^2/2 (padded to 1 KiB)
    static constexpr int SUB = 2;                      // tiles per barrier interval
    static constexpr int SUPER_B = SUB * TILE_B;
    static constexpr int NCHUNK = SUPER_B / 1024;
    static constexpr int PMAX = (NCHUNK + W - 1) / W;  // 1-KiB pieces per wave and interval
    static constexpr int BUF_B = PMAX * W * 1024;      // LDS bytes per buffer (every wave copies PMAX pieces, unconditionally)
    static constexpr int NB = 2;                       // buffers: interval g + 1 is requested at the start of interval g
    static constexpr int SMEM = NB * BUF_B;
};

template <int DT, int METRIC, bool HASQ>
__global__ void __launch_bounds__(256, 2) vq_screenc_kernel(const ScreenArgs a, const int nblk, long long *const trace)
{
#ifdef VQC_TRACE            // dev build: s_memtime stamps of the first VQC_TRACE_IV intervals of workgroups 0 .. 7 (tools/trace_screenc.py)
#define VQC_TRACE_IV 64
    // workgroups 0 .. 7 and (their likely CU partners) 256 .. 263
    const int trace_wg = blockIdx.x < 8 ? (int)blockIdx.x : (blockIdx.x >= 256 && blockIdx.x < 264 ? (int)blockIdx.x - 248 : -1);
#define VQC_STAMP(k) do { if (trace && trace_wg >= 0 && g < VQC_TRACE_IV && lane == 0) \
        trace[(((size_t)trace_wg * 4 + wave) * VQC_TRACE_IV + g) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define VQC_STAMP(k) do {} while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int lcnt[2];                            // this workgroup's list segment: open / pair entries so far
    using Cfg = ScreenCCfg<DT>;
    constexpr int NK = Cfg::NK, TILE_B = Cfg::TILE_B, SUB = Cfg::SUB, SUPER_B = Cfg::SUPER_B;
    constexpr int PMAX = Cfg::PMAX, BUF_B = Cfg::BUF_B, NB = Cfg::NB;
#ifndef VQC_PF
#define VQC_PF 3
#endif
    constexpr int PF = VQC_PF < NK ? VQC_PF : NK;      // A-fragment ring depth
    static_assert(SUB == 2, "an interval is two tiles: the accumulator sets swap roles inside it");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave;
    const int j = lane & 31;
    const int half = lane >> 5;
    const int G2 = (int)gridDim.x;                     // workgroups in the launch
    const int gid = (int)blockIdx.x;                   // this workgroup takes the 256-row blocks gid, gid + G2, ...
    const int nt16 = a.n_tiles16;
    const int nst = nt16 / SUB;
    const char *const tiles = a.tiles16;
    const int sc = (int)a.scalars[2];
    const float rho = __uint_as_float(a.scalars[1]);
    const float r0 = __uint_as_float(a.scalars[3]);
    const bool y2ok = __uint_as_float(a.scalars[0]) < 1e37f;
    // PLAIN mode (chosen per launch from the codebook alone): while the largest code norm is within 4 x of the smallest, charging every
    // code OTHER than the two best with the largest norm costs little -- the winner / runner-up decision keeps both codes' own
    // allowances, only the "nobody else comes close" test (best vs third) is looser, i.e. some pair rows become open rows -- and the
    // sweep can track the plain scores: one start vector -||c||^2 / 2 for both row blocks and all lanes, 8 packed multiplies per tile
    // instead of 8 + 16 packed FMAs (which cost the cfg-2 search 4 %).  Codebooks with a wider norm spread keep the upper-bound scores
    // (one large code must not raise every row's threshold).
    const float y2max_c = __uint_as_float(a.scalars[0]), y2min_c = __uint_as_float(~a.scalars[4]);
#ifdef VQC_FORCE_UPPER      // dev builds: the upper-bound sweep whatever the codebook (A/B runs of the two sweeps on one workload)
    const bool plain = false;
#else
    const bool plain = __builtin_amdgcn_readfirstlane((y2ok && y2min_c > 0.f && y2max_c <= 16.f * y2min_c) ? 1 : 0) != 0;
#endif
    const float ymax_c = sqrtf(y2max_c) * 1.0001f;
    const int ldq2 = (int)(a.ldq * 2);
    const int ldx2 = (int)(a.ldx * 2);
    const int nb_mine = gid < nblk ? (nblk - gid + G2 - 1) / G2 : 0;

    // ---- small helpers ------------------------------------------------------------------------------------------------------
    auto pick_sx = [&](unsigned mx) {                  // as in vq_screen16_kernel: largest finite ||x||^2 (float bits) -> norm below 2^14
        const int e2 = (int)(mx >> 23) - 127;
        int SX = (mx == 0u) ? 0 : 14 - ((e2 >> 1) + 1);
        SX = SX > 120 - sc ? 120 - sc : SX;
        SX = SX < -120 - sc ? -120 - sc : SX;
        SX = SX > sc + 90 ? sc + 90 : SX;
        return SX > 126 ? 126 : (SX < -126 ? -126 : SX);
    };
    // the certificate's per-code pieces (vq_screen.hip header, round 6): e(c) = Rrow + Arow ||c|| + kb ||c||^2; one operand set, bf16 rows
    auto arow_of = [&](float xs, int SX) {
        const float u = 5.9604645e-8f;
        const float conv = u * sqrtf((float)DT) * __uint_as_float((unsigned)(127 - SX) << 23);
        const float xn = sqrtf(xs) * 1.0001f;
        const float nacc = (float)(DT + 1);
        const float slack = 1.0001f + 4.f * nacc * u;      // (+ the MFMA's rounding of the added part itself)
        if (METRIC == 0) return (xn * (u * (10.f + (float)DT + 2.002f * nacc) + rho) + conv) * slack;
        return (xn * (u * (float)(DT + 2 * DT) * 1.001f + rho) + conv) * slack;
    };
    auto rrow_of = [&](float xs) {
        const float xn = sqrtf(xs) * 1.0001f;
        if (METRIC == 0) return 5.f * 5.9604645e-8f * xs + xn * r0 + 2e-8f;
        return xn * r0 + 1e-30f;
    };
    auto conv_word = [&](unsigned w, float S) -> unsigned {   // two bf16 -> two fp16, exact above 2^-14 (scaled), truncated below
        const float lo = __uint_as_float(w << 16) * S, hi = __uint_as_float(w & 0xffff0000u) * S;
        return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
    };
    auto finite_bits = [&](float v) -> unsigned {
        const unsigned bits = __float_as_uint(v);
        return (bits & 0x7f800000u) == 0x7f800000u ? 0u : (bits & 0x7fffffffu);
    };
    auto xor32 = [&](unsigned v, int half) -> unsigned {   // the value of lane ^ 32 (v_permlane32_swap: no index register, no LDS)
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return half ? r[0] : r[1];
    };
    auto xor32f = [&](float v, int half) -> float { return __uint_as_float(xor32(__float_as_uint(v), half)); };
    // lane number, re-derived where it is needed (prologue, breaks): held in a register across the sweep, it and everything hipcc
    // derives from it ahead of time (row offsets, 16 j, ...) is spilled to scratch around the sweep loop, and every reload in
    // the break is a serialised trip to memory
    auto lane_now = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto wave_max_u = [&](unsigned v) -> unsigned {     // DPP inside the rows of 16 + four readlanes
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
        const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
        const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
        return max(max(r0, r1), max(r2, r3));
    };
    auto rank_in = [&](unsigned long long mask) -> int {   // set bits of `mask` below this lane
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    };
    // merge of two descending triples with the codes of their first two entries (a: this side, p: the other side)
    auto merge3 = [&](float a1, float a2, float a3, int ia1, int ia2, float p1, float p2, float p3, int ib1, int ib2,
                      float &c1, float &c2, float &c3, int &ic1, int &ic2) __attribute__((always_inline)) {
        const bool take = p1 > a1;
        const float h1 = take ? p1 : a1, h2 = take ? p2 : a2, h3 = take ? p3 : a3;   // the triple that holds the best
        const float l1 = take ? a1 : p1, l2 = take ? a2 : p2;                          // the other one
        const int ih1 = take ? ib1 : ia1, ih2 = take ? ib2 : ia2, il1 = take ? ia1 : ib1;
        const bool second_low = l1 > h2;                                               // runner-up comes from the other triple
        c1 = h1; ic1 = ih1;
        c2 = second_low ? l1 : h2;
        ic2 = second_low ? il1 : ih2;
        c3 = second_low ? fmaxf(h2, l2) : fmaxf(h3, l1);
    };

    // ---- state ---------------------------------------------------------------------------------------------------------------
    // lane (j, half) of a wave: row j of both of the wave's 32-row blocks, elements 16 ks + 8 half + 0..7 of every k-step
    uint4 cur[2][NK];                                  // B operands of the block being swept (raw bf16 rows between request and conversion)
    f32x16 accA[2], accB[2];                           // two accumulator sets x two row blocks
    float m1[2], m2[2], m3[2];                         // top 3 per row block
    int tix[2], tix2[2];
    float arow_c[2], rrow_c[2], SS_c = 1.f, iSS_c = 1.f;      // Arow (unscaled), Rrow per row block
    int g = 0;                                         // interval counter of the workgroup
    int bcur = 0;                                      // LDS buffer of interval g
    int tpc = 0;                                       // g mod nst: the tile pair of interval g
    int blk = gid;                                     // this group's current 256-row block

    // ---- the tile stream --------------------------------------------------------------------------------------------------------
    // Interval g + 1 is requested at the start of interval g by all 8 waves (LDS-DMA, 1 KiB per instruction, lane-linear image =
    // the packed tiles' own layout; as asm: hipcc drains every DMA it knows of before the next ds_read) into the other buffer, last
    // read in interval g - 1: everybody has finished that interval, or they would not have passed the barrier that ends it.
    auto issue_stage = [&]() __attribute__((always_inline)) {
        unsigned lane16;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(lane16));
        const int tp = tpc + 1 == nst ? 0 : tpc + 1;
        const int bnext = bcur + 1 == NB ? 0 : bcur + 1;
        const unsigned wc = (unsigned)wave * (unsigned)(PMAX * 1024) + lane16;
        const char *const gp = tiles + (size_t)tp * SUPER_B;
        const unsigned lb = (unsigned)(uintptr_t)(smem + bnext * BUF_B + wave * (PMAX * 1024));
        static_assert(PMAX <= 12, "groups of up to four pieces");
#pragma unroll
        for (int i0 = 0; i0 + 4 <= PMAX; i0 += 4)     // (the instruction offset moves source and destination alike)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072\n\t"
                         :: "v"(wc), "s"(gp + i0 * 1024), "s"(lb + (unsigned)i0 * 1024u) : "memory");
#pragma unroll
        for (int i = PMAX / 4 * 4; i < PMAX; ++i)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                         :: "v"(wc), "s"(gp + i * 1024), "s"(lb + (unsigned)i * 1024u) : "memory");
    };
    // end of an interval: this wave's pieces have landed (YOUNGER = memory instructions this wave has issued after them in this
    // interval and does not wait for here; vmcnt retires in order), then the barrier
#define VQC_END_INTERVAL(YOUNGER) do { \
        VQC_STAMP(1); \
        asm volatile("s_waitcnt vmcnt(" #YOUNGER ")\n\ts_barrier" ::: "memory"); \
        VQC_STAMP(2); \
        bcur = bcur + 1 == NB ? 0 : bcur + 1; tpc = tpc + 1 == nst ? 0 : tpc + 1; ++g; (void)g; } while (0)

    // the rows of block `b`: wave-uniform base (SGPRs) + 32-bit lane offset; rows past the end repeat the last one
    auto load_rows = [&](int b, int j, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int64_t r0 = (int64_t)b * 256 + wq * 64 + rb * 32;
            const int64_t lim = a.N - 1 - r0;
            const char *const xb = (const char *)a.x + (r0 < a.N ? r0 : a.N - 1) * a.ldx * 2;
            unsigned xo = (unsigned)min(j, lim > 31 ? 31 : (lim < 0 ? 0 : (int)lim)) * (unsigned)ldx2 + (unsigned)half * 16u;
            asm volatile("" : "+v"(xo));
            const char *const xl = xb + xo;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) cur[rb][ks] = *(const uint4 *)(xl + ks * 32);
        }
    };
    // raw rows -> norms, the wave's scale, fp16 operands in place, thresholds
    auto convert_rows = [&](int half) __attribute__((always_inline)) {
        float xs2[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float xs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const unsigned w[4] = {cur[rb][ks].x, cur[rb][ks].y, cur[rb][ks].z, cur[rb][ks].w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xs[q] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[q]), __builtin_bit_cast(bf16x2, w[q]), xs[q], false);
            }
            float t = (xs[0] + xs[1]) + (xs[2] + xs[3]);
            t += xor32f(t, half);
            xs2[rb] = t * 1.001f;
        }
        const int SX = pick_sx(wave_max_u(max(finite_bits(xs2[0]), finite_bits(xs2[1]))));
        const float S = __uint_as_float((unsigned)(SX + 127) << 23);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
            {
                cur[rb][ks] = make_uint4(conv_word(cur[rb][ks].x, S), conv_word(cur[rb][ks].y, S), conv_word(cur[rb][ks].z, S), conv_word(cur[rb][ks].w, S));
                // pinned HERE (the consumer is the next sweep's first MFMA: hipcc sinks the conversion behind the interval's barrier)
                asm volatile("" : "+v"(cur[rb][ks].x), "+v"(cur[rb][ks].y), "+v"(cur[rb][ks].z), "+v"(cur[rb][ks].w));
            }
        SS_c = __uint_as_float((unsigned)(SX + sc + 127) << 23);
        iSS_c = __uint_as_float((unsigned)(127 - SX - sc) << 23);
        // per ROW (a wave-uniform factor was measured: ~1 % of the cfg-2 search against 55 % instead of 7 % open rows when the row norms
        // of a batch span 30 x)
        arow_c[0] = arow_of(xs2[0], SX); arow_c[1] = arow_of(xs2[1], SX);
        rrow_c[0] = rrow_of(xs2[0]); rrow_c[1] = rrow_of(xs2[1]);
#pragma unroll
        for (int t = 0; t < 2; ++t)     // overflow guard (vq_screen16_kernel): such a row is never certified and adds nothing to the start values
        {   // (selects, not a branch: as `if (...) { ... }` this guard made hipcc spill 84 registers around the sweep)
            const bool ok = (arow_c[t] < 1e30f) & (arow_c[t] * SS_c < 1e30f) & (rrow_c[t] < 1e30f) & y2ok;
            arow_c[t] = ok ? arow_c[t] : 0.f;
            rrow_c[t] = ok ? rrow_c[t] : __builtin_inff();
        }
    };
    auto reset_fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) { m1[t] = m2[t] = m3[t] = -__builtin_inff(); tix[t] = tix2[t] = 0; }
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[0][r] = accB[1][r] = -3.0e38f;   // "padding codes" in front of the first tile: never win
    };

    // one score into (best, second, third): the key carries the register number in its 4 low mantissa bits; ONE asm statement per
    // score (as builtins hipcc sinks every v_med3 of a tile behind the tile's last MFMA, vq_screen.hip)
    auto fold = [&](const f32x16 &acc, int e, float &b1, float &b2, float &b3) __attribute__((always_inline)) {
        const float k = __uint_as_float((__float_as_uint(acc[e]) & 0xfffffff0u) | (unsigned)e);
        asm volatile("v_med3_f32 %2, %1, %2, %3\n\tv_med3_f32 %1, %0, %1, %3\n\tv_max_f32 %0, %0, %3"
                     : "+v"(b1), "+v"(b2), "+v"(b3) : "v"(k));
    };
    auto book = [&](int t, float om1, float om2, int tile_id) __attribute__((always_inline)) {   // which tiles hold best / second of block t
        const bool c1 = m1[t] != om1;
        const int from_old_best = (c1 && m2[t] == om1) ? tix[t] : tile_id;
        tix2[t] = (m2[t] != om2) ? from_old_best : tix2[t];
        tix[t] = c1 ? tile_id : tix[t];
        asm volatile("" : "+v"(tix[t]), "+v"(tix2[t]));
    };

    // ---- start offsets ----------------------------------------------------------------------------------------------------------
#ifndef VQC_HALF
#define VQC_HALF 40         // x 1024 cycles: about half a period
#endif
#ifndef VQC_STAGGER
#define VQC_STAGGER 0       // x 1024 cycles, times the CU's phase 0 .. 31
#endif
    {
        // the workgroup whose LDS allocation does not start at 0 is the second one on its CU (HW_REG_LDS_ALLOC, LDS_BASE field)
        const bool second_wg = (__builtin_amdgcn_s_getreg((11 << 11) | (0 << 6) | 6) & 0xfff) != 0;
        const long long wait = 1024ll * ((second_wg ? VQC_HALF : 0) + (((int)blockIdx.x >> 3) & 31) * VQC_STAGGER);
        const long long t0 = __builtin_readcyclecounter();
        while ((long long)__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#ifdef VQC_TRACE
    if (trace && tid == 0) {    // where every workgroup runs: HW_ID (id 4), XCC_ID (id 20), LDS_ALLOC (id 6), start time
        long long *const w = trace + 16 * 4 * VQC_TRACE_IV * 8 + (size_t)blockIdx.x * 4;
        w[0] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        w[1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
        w[2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);
        w[3] = __builtin_readcyclecounter();
    }
#endif
    // ---- prologue: interval 0's tiles (buffer 0), the first block's rows ---------------------------------------------------------
    {
        const unsigned wc = (unsigned)wave * (unsigned)(PMAX * 1024) + (unsigned)lane * 16u;
#pragma unroll
        for (int k = 0; k < PMAX; ++k)
            *(f32x4 *)(smem + wc + k * 1024) = *(const f32x4 *)(tiles + wc + (size_t)k * 1024);
        if (tid < 2) lcnt[tid] = 0;
        load_rows(blk < nblk ? blk : nblk - 1, j, half);
        convert_rows(half);
        reset_fold();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int ptile = 0;
    for (int ib = 0; ib < nb_mine; ++ib, blk += G2) {
        // ---- sweep: nst intervals of two tiles, starting wherever the tile cycle stands; one copy of the loop per certificate mode ----
        auto sweep = [&](auto plain_c) __attribute__((always_inline)) {
        constexpr bool PLAIN = decltype(plain_c)::value;
#pragma unroll 1
        for (int iv = 0; iv < nst; ++iv) {
            VQC_STAMP(0);
            issue_stage();
            const char *const sbase = smem + bcur * BUF_B;
            const int tp = tpc;
            unsigned lane16;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(lane16));
            // one tile: accumulate into C, fold the previous tile's P
            auto tile_body = [&](int sub, f32x16 (&C)[2], const f32x16 (&P)[2]) __attribute__((always_inline)) {
                const char *tile = sbase + sub * TILE_B;
                const float *nh = (const float *)(tile + 64 * DT + ((lane16 >> 9) << 4));   // + 4 * half floats
                const int tile_id = tp * SUB + sub;
                const bool has_pad = (tile_id + 1) * 32 > a.C;
                const uint4 *ap = (const uint4 *)(tile + lane16);
                const float o10 = m1[0], o20 = m2[0], o11 = m1[1], o21 = m2[1];
                uint4 af[PF];
#pragma unroll
                for (int p = 0; p < PF; ++p) af[p] = ap[p * 64];
                // start value -||c||^2 / 2 (scaled); register e <-> code 8 (e >> 2) + 4 half + (e & 3) of the tile.  Tiles with padding
                // codes clamp it to a finite -3e38.
                // + (row factor) x ||c|| (tile tail, floats 32 ..): every tracked score is an upper bound of the code's true score
                f32x16 init0, init1;
                if constexpr (PLAIN) {      // the plain start values (tile tail, floats 64 ..), shared by both row blocks
                    if (METRIC == 0 || has_pad) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v = *(const f32x4 *)(nh + 64 + 8 * q);
                            if (METRIC != 0) { v.x = v.x < -1e38f ? v.x : 0.f; v.y = v.y < -1e38f ? v.y : 0.f;
                                               v.z = v.z < -1e38f ? v.z : 0.f; v.w = v.w < -1e38f ? v.w : 0.f; }
                            const f32x2 r0 = f32x2{v.x, v.y} * f32x2{SS_c, SS_c}, r1 = f32x2{v.z, v.w} * f32x2{SS_c, SS_c};
                            init0[4 * q + 0] = r0.x; init0[4 * q + 1] = r0.y; init0[4 * q + 2] = r1.x; init0[4 * q + 3] = r1.y;
                        }
                        if (has_pad) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) init0[r] = fmaxf(init0[r], -3.0e38f);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) init0[r] = 0.f;
                    }
                } else {
                    const float as0 = arow_c[0] * SS_c, as1 = arow_c[1] * SS_c;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v = *(const f32x4 *)(nh + 8 * q);
                        const f32x4 w = *(const f32x4 *)(nh + 32 + 8 * q);
                        if (METRIC != 0) { v.x = v.x < -1e38f ? v.x : 0.f; v.y = v.y < -1e38f ? v.y : 0.f;
                                           v.z = v.z < -1e38f ? v.z : 0.f; v.w = v.w < -1e38f ? v.w : 0.f; }
                        // two scores per instruction (v_pk_mul_f32 / v_pk_fma_f32, explicit 2-vectors: the SLP vectoriser is off)
                        const f32x2 vp[2] = {f32x2{v.x, v.y}, f32x2{v.z, v.w}}, wp[2] = {f32x2{w.x, w.y}, f32x2{w.z, w.w}};
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            f32x2 r0, r1;
                            if (METRIC == 0 || has_pad) {
                                const f32x2 b = vp[i] * f32x2{SS_c, SS_c};
                                r0 = __builtin_elementwise_fma(wp[i], f32x2{as0, as0}, b);
                                r1 = __builtin_elementwise_fma(wp[i], f32x2{as1, as1}, b);
                            } else {
                                r0 = wp[i] * f32x2{as0, as0};
                                r1 = wp[i] * f32x2{as1, as1};
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
                    const f16x8 av = __builtin_bit_cast(f16x8, af[s % PF]);
                    C[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, cur[0][s]), s == 0 ? init0 : C[0], 0, 0, 0);
                    C[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, cur[1][s]), s == 0 ? (PLAIN ? init0 : init1) : C[1], 0, 0, 0);
                    if (s + PF < NK) af[s % PF] = ap[(s + PF) * 64];
#ifndef VQC_NO_FOLD
                    fold(P[0], s, m1[0], m2[0], m3[0]);
                    fold(P[1], s, m1[1], m2[1], m3[1]);
#endif
                    __builtin_amdgcn_sched_barrier(0);   // pins the slice and the prefetch distance between the MFMAs
                }
                book(0, o10, o20, ptile);
                book(1, o11, o21, ptile);
                ptile = tile_id;
            };
            tile_body(0, accA, accB);
            tile_body(1, accB, accA);
            VQC_END_INTERVAL(0);
        }
        };
        if (plain) sweep(std::true_type{}); else sweep(std::false_type{});

        // ---- break interval 0: last tile's fold, classification, index, list slots, q rows, list entries; the next block's rows are requested ----
        VQC_STAMP(0);
        issue_stage();
        const int lane_b = lane_now();
        const int j_b = lane_b & 31, half_b = lane_b >> 5;
        const unsigned j16_b = (unsigned)j_b * 16u;
        const bool has_next = ib + 1 < nb_mine;
        {
            const float o10 = m1[0], o20 = m2[0], o11 = m1[1], o21 = m2[1];
#pragma unroll
            for (int e = 0; e < 16; ++e) { fold(accB[0], e, m1[0], m2[0], m3[0]); fold(accB[1], e, m1[1], m2[1], m3[1]); }
            book(0, o10, o20, ptile);
            book(1, o11, o21, ptile);
        }
        VQC_STAMP(3);
        // merge the half-waves, classify (certified / pair / open), emit the index; lane l then carries row l of the wave's 64
        int code, cls, id2, prow;
        {
            int codes[2], id2s[2];
            bool cert[2], pairf[2];
            float dbg4[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int e1 = (int)(__float_as_uint(m1[t]) & 15u), e2b = (int)(__float_as_uint(m2[t]) & 15u);
                const int ia1 = tix[t] * 32 + 8 * (e1 >> 2) + 4 * half_b + (e1 & 3);
                const int ia2 = tix2[t] * 32 + 8 * (e2b >> 2) + 4 * half_b + (e2b & 3);
                float b1, b2, b3;
                merge3(m1[t], m2[t], m3[t], ia1, ia2, xor32f(m1[t], half_b), xor32f(m2[t], half_b), xor32f(m3[t], half_b), (int)xor32((unsigned)ia1, half_b), (int)xor32((unsigned)ia2, half_b),
                       b1, b2, b3, codes[t], id2s[t]);
                // the winner's own code part A ||c|| + kb ||c||^2 (||c|| from the tile tail of the packed codebook)
                const float kb = 5.9604645e-8f * (5.f + 1.001f * (float)(DT + 1) + 0.51f) * 1.001f;
                auto code_part = [&](int c) {
                    const int cc = c < a.C ? c : 0;
                    const float yb = *(const float *)(tiles + (size_t)(cc >> 5) * TILE_B + 64 * DT + (32 + (cc & 31)) * 4);
                    return (arow_c[t] * yb + (METRIC == 0 ? kb * yb * yb * 1.001f : 0.f)) * 1.0001f;
                };
                const float cp1 = code_part(codes[t]);
                float thr = (2.f * cp1 + 2.f * rrow_c[t]) * SS_c + 8e-6f * fabsf(b1);
                if (plain) {
                    // plain scores: the winner's and the runner-up's own allowances between the two, the winner's + the LARGEST code's
                    // against everything else (third and beyond)
                    const float cp2 = code_part(id2s[t]);
                    const float cpm = (arow_c[t] * ymax_c + (METRIC == 0 ? kb * ymax_c * ymax_c * 1.001f : 0.f)) * 1.0001f;
                    thr = (cp1 + cp2 + 2.f * rrow_c[t]) * SS_c + 8e-6f * fabsf(b1);
                    const float thr3 = (cp1 + cpm + 2.f * rrow_c[t]) * SS_c + 8e-6f * fabsf(b1);
                    const bool others_out = (b1 - b3) > thr3;
                    cert[t] = others_out && ((b1 - b2) > thr) && codes[t] < a.C && b1 < 3.0e38f;
                    pairf[t] = !cert[t] && others_out && codes[t] < a.C && id2s[t] < a.C && b1 < 3.0e38f;
                } else {
                    cert[t] = ((b1 - b2) > thr) && codes[t] < a.C && b1 < 3.0e38f;
                    pairf[t] = !cert[t] && ((b1 - b3) > thr) && codes[t] < a.C && id2s[t] < a.C && b1 < 3.0e38f;
                }
                dbg4[t][0] = dbg4[t][1] = dbg4[t][2] = 0.f;
                if (a.dbg && plain) { dbg4[t][0] = b1 * iSS_c; dbg4[t][1] = b2 * iSS_c; dbg4[t][2] = thr * iSS_c; }
                if (a.dbg && !plain) {    // debug view in the units of t = x.c - ||c||^2 / 2 (vq_screen16_kernel): what the sweep added is taken off again
                    float add[2], sc2[2];
                    const int cs[2] = {codes[t], id2s[t]};
                    const float us[2] = {b1, b2};
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int cc = cs[k] < a.C ? cs[k] : 0;
                        const float *tl = (const float *)(tiles + (size_t)(cc >> 5) * TILE_B + 64 * DT) + (cc & 31);
                        const float v = METRIC == 0 ? tl[0] : 0.f, w = tl[32], pz = METRIC == 0 ? tl[64] : 0.f;
                        const float init = METRIC == 0 ? __builtin_fmaf(w, arow_c[t] * SS_c, v * SS_c) : w * (arow_c[t] * SS_c);
                        sc2[k] = (us[k] - init) * iSS_c + pz;
                        add[k] = init * iSS_c - pz;
                    }
                    dbg4[t][0] = sc2[0]; dbg4[t][1] = sc2[1]; dbg4[t][2] = thr * iSS_c - add[0] + add[1];
                }
            }
            code = half_b ? codes[1] : codes[0];
            id2 = half_b ? id2s[1] : id2s[0];
            const bool certified = half_b ? cert[1] : cert[0];
            const bool pair = half_b ? pairf[1] : pairf[0];
            const int64_t row = (int64_t)blk * 256 + wq * 64 + lane_b;
            const bool on = row < a.N;
            if (on) {
                a.idx_out[row * a.idx_stride] = (int64_t)(code < a.C ? code : 0);
                if (a.dbg) {
                    float *d = a.dbg + row * 4;
                    d[0] = half_b ? dbg4[1][0] : dbg4[0][0]; d[1] = half_b ? dbg4[1][1] : dbg4[0][1]; d[2] = half_b ? dbg4[1][2] : dbg4[0][2];
                    d[3] = certified ? 0.f : (pair ? 2.f : 1.f);
                }
            }
            if (code >= a.C) code = 0;
            cls = !on ? 0 : (certified ? 0 : (pair ? 2 : 1));
            prow = (int)row;
        }
        const unsigned long long balo = __ballot(cls == 1), balp = __ballot(cls == 2);
        // list space: this workgroup's own segment of the staging lists (open rows from its front, pair rows from its back), handed
        // out by two LDS counters.  One global counter pair for the whole launch would be the launch's bottleneck: same-address
        // atomics retire at about one per 45 cycles on this chip, and a launch appends from 16 k waves (N = 2^20).
        int base_o = 0, base_p = 0;
        if (lane_b == 0) {
            if (balo) base_o = atomicAdd(&lcnt[0], (int)__popcll(balo));
            if (balp) base_p = atomicAdd(&lcnt[1], (int)__popcll(balp));
        }
        const int64_t prow0 = (int64_t)blk * 256 + wq * 64;
        const int64_t plim64 = a.N - 1 - prow0;
        const int plim = plim64 > 63 ? 63 : (plim64 < 0 ? 0 : (int)plim64);
        char *const pqbase = (char *)a.q_out + (prow0 < a.N ? prow0 : a.N - 1) * a.ldq * 2;
        VQC_STAMP(4);
        // q rows (gather of bf16 code rows from L2 -> store, two rows per instruction): 16 row pairs in flight in the accumulators'
        // registers (free until the next sweep) -- two round trips to L2 for the 64 rows
#ifndef VQC_QB
#define VQC_QB 16
#endif
        constexpr int QB = VQC_QB;                         // row pairs in flight
        if (HASQ) {
#pragma unroll
            for (int t0 = 0; t0 < 32; t0 += QB) {
                u32x4 gq[QB];
#pragma unroll
                for (int t = 0; t < QB; ++t) {
                    const int c = __builtin_amdgcn_ds_bpermute((2 * (t0 + t) + half_b) * 4, code);
                    gq[t] = *(const u32x4 *)((const char *)a.embed_bf16 + ((unsigned)c * (unsigned)(DT * 2) + j16_b));
                }
#pragma unroll
                for (int t = 0; t < QB; ++t) {
                    const int rl = min(2 * (t0 + t) + half_b, plim);
                    *(u32x4 *)(pqbase + ((unsigned)(rl * ldq2) + j16_b)) = gq[t];
                }
            }
        }
        VQC_STAMP(5);
        // list entries
        {
            const size_t seg0 = (size_t)blockIdx.x * (size_t)a.seg_cap;
            if (balo) {
                const int bo = __builtin_amdgcn_readfirstlane(base_o);
                if (cls == 1) {
                    const size_t slot = seg0 + (size_t)(bo + rank_in(balo));
                    a.seg_rows[slot] = prow;
                    a.seg_keys[slot] = ~0ull;
                }
            }
            if (balp) {
                const int bp = __builtin_amdgcn_readfirstlane(base_p);
                if (cls == 2) {
                    const size_t slot = seg0 + (size_t)(a.seg_cap - 1 - (bp + rank_in(balp)));
                    a.seg_rows[slot] = prow;
                    a.seg_keys[slot] = (unsigned long long)(unsigned)code | ((unsigned long long)(unsigned)id2 << 32);
                }
            }
        }
        VQC_STAMP(6);
        // the next block's rows are requested LAST (until here the operand registers carried the q rows): everything but these 32
        // loads and the q rows' last stores has retired when the interval ends
        __builtin_amdgcn_sched_barrier(0);
        load_rows(has_next ? blk + G2 : blk, j_b, half_b);
        __builtin_amdgcn_sched_barrier(0);
        VQC_STAMP(7);
        if (HASQ) VQC_END_INTERVAL(63); else VQC_END_INTERVAL(32);

        // ---- break interval 1: nothing but the tile stream: the rows are on their way ----
        VQC_STAMP(0);
        issue_stage();
        VQC_END_INTERVAL(0);

        // ---- break interval 2: the next block's rows have arrived: norms, scale, conversion, thresholds ----
        VQC_STAMP(0);
        issue_stage();
        convert_rows(half_b);
        reset_fold();
        VQC_END_INTERVAL(0);
    }
    __syncthreads();
    if (tid < 2) a.seg_counts[2 * blockIdx.x + tid] = lcnt[tid];
}

// packs the list segments of vq_screenc_kernel into the lists the exact passes read: open rows flag_rows / flag_keys [0, n_open), pair
// rows from the back [N - 1 - k], and the two totals into flag_count.  One workgroup per segment; the 2 x nseg counters are summed by
// every workgroup for itself (nseg <= 512).
__global__ void __launch_bounds__(256) vq_compact_lists_kernel(const ScreenArgs a, const int nseg)
{
    __shared__ int red[2][4];
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int po = 0, pp = 0;                                // entries of the segments before this one
    for (int s = tid; s < seg; s += 256) { po += a.seg_counts[2 * s]; pp += a.seg_counts[2 * s + 1]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { po += __shfl_xor(po, o, 64); pp += __shfl_xor(pp, o, 64); }
    if (lane == 0) { red[0][wave] = po; red[1][wave] = pp; }
    __syncthreads();
    po = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    pp = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const int no = a.seg_counts[2 * seg], np = a.seg_counts[2 * seg + 1];
    const size_t seg0 = (size_t)seg * (size_t)a.seg_cap;
    for (int i = tid; i < no; i += 256) {
        a.flag_rows[po + i] = a.seg_rows[seg0 + i];
        a.flag_keys[po + i] = a.seg_keys[seg0 + i];
    }
    for (int i = tid; i < np; i += 256) {
        const size_t src = seg0 + (size_t)(a.seg_cap - 1 - i);
        const int64_t dst = a.N - 1 - (pp + i);
        a.flag_rows[dst] = a.seg_rows[src];
        a.flag_keys[dst] = a.seg_keys[src];
    }
    if (seg == nseg - 1 && tid == 0) { a.flag_count[0] = po + no; a.flag_count[1] = pp + np; }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
#ifndef VQC_MIN_ROWS
#define VQC_MIN_ROWS (512 * 256)      // below one block per workgroup the 4-wave kernel spreads the rows better
#endif

static long long *vqc_g_trace = nullptr;              // dev builds with -DVQC_TRACE: where the kernel puts its s_memtime stamps
extern "C" void vqhip_screenc_set_trace(long long *p) { vqc_g_trace = p; }

static int vqc_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("VQHIP_SCREEN_PERSIST"); v = (e && e[0] == '0') ? 0 : 1; }   // =0: the 4-wave kernel (A/B runs)
    return v;
}

int vq_screenc_eligible(const ScreenArgs &a, int x_dtype, int DT)
{
    if (!vqc_enabled()) return 0;
    if (x_dtype != VQHIP_BF16 || DT != 256 || a.heads > 1) return 0;
    if (a.n_tiles16 < 2 || (a.n_tiles16 & 1)) return 0;
    if (a.resid_out || a.sqerr_partial || a.prev_idx) return 0;
    if (a.N < VQC_MIN_ROWS) return 0;
    if (a.q_out && ((((uintptr_t)a.q_out) & 15) || ((a.ldq * 2) & 15))) return 0;
    return 1;
}

template <int METRIC, bool HASQ>
static int vqc_launch(const ScreenArgs &a, hipStream_t st)
{
    using Cfg = ScreenCCfg<256>;
    static VqAttrOnce once;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
        else n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (int rc = vq_set_max_smem(once, (const void *)vq_screenc_kernel<256, METRIC, HASQ>, Cfg::SMEM, "vq_screenc_kernel")) return rc;
    const int nblk = (int)((a.N + 255) / 256);
    int grid = nblk < 2 * n_cu ? nblk : 2 * n_cu;
    if (grid > VQ_SEG_MAX) grid = VQ_SEG_MAX;
    ScreenArgs b = a;
    b.seg_cap = (nblk + grid - 1) / grid * 256;        // rows of the busiest workgroup
    hipLaunchKernelGGL((vq_screenc_kernel<256, METRIC, HASQ>), dim3((unsigned)grid), dim3(256), Cfg::SMEM, st, b, nblk, vqc_g_trace);
    if (int rc = vq_launch_status("vq_screenc_kernel")) return rc;
    hipLaunchKernelGGL(vq_compact_lists_kernel, dim3((unsigned)grid), dim3(256), 0, st, b, grid);
    return vq_launch_status("vq_compact_lists_kernel");
}

int vq_screenc_launch(const ScreenArgs &a, int metric_is_cosine, hipStream_t st)
{
    if (a.q_out) return metric_is_cosine ? vqc_launch<1, true>(a, st) : vqc_launch<0, true>(a, st);
    return metric_is_cosine ? vqc_launch<1, false>(a, st) : vqc_launch<0, false>(a, st);
}
