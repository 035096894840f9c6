// vq_screen_p.hip -- persistent, software-pipelined form of the single-pass fp16 screen (vq_screen16_kernel in vq_screen.hip),
// gfx950 only.  Same certificate, same outputs (index, q rows, open / pair lists); what changes is WHEN things happen:
//
//   * ONE workgroup of 8 waves per CU (two waves per SIMD), each wave owns ONE 32-row block at a time and loops over row blocks
//     ("super-block" sb = 256 consecutive rows per workgroup; workgroup b takes sb = b, b + G, b + 2 G, ...).
//   * The memory phases that vq_screen16_kernel runs before and after its sweep are issued BETWEEN the MFMAs of the sweep:
//       - the next row block's raw bf16 rows are requested during the first tiles of the sweep (64 registers), their norms are
//         summed, the scale agreed and the rows converted to fp16 operands in place while the current block is swept;
//       - the previous row block's q rows (gather of bf16 code rows from L2 -> store) are moved two rows per instruction pair
//         during the first 16 tiles.
//     The order of the memory instructions inside a tile is fixed (staging loads, x loads, q gather ... staging stores, q store)
//     so that every wait the compiler inserts is a counted vmcnt(n) on an OLDER short-latency load, never on the HBM loads.
//   * The row scale 2^SX is agreed per WORKGROUP iteration (largest finite row norm of the 256 rows, exchanged through LDS), so
//     the start values -||c||^2/2 * 2^(SX+sc) are scaled ONCE per iteration into an LDS copy and enter the accumulator through
//     the MFMA's C operand: no multiply per score register and tile (the 4-wave kernel spends 16 v_mul per tile on it).
//     fp16 keeps 29 binades of normal range: rows within 2^-20 of the largest norm of their super-block lose nothing; what
//     is lost below that is charged by `conv` in the bound exactly as before.
//
// Eligibility (launch_screen in vq_screen.hip): bf16 rows, D = 256, at least 32 fp16 tiles (C > 992), at most VQP_MAX_NORM_CODES
// codes, no residual / squared-error output, N >= VQP_MIN_ROWS.  Everything else keeps vq_screen16_kernel.
//
// Reference arithmetic that the screen certifies: cdist at vqp.py:58-62, argmax at vqp.py:140 (cosine: einsum at vqp.py:741).

#include <type_traits>
#include <utility>

#include "vq_screen_args.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define VQP_MAX_NORM_CODES 8192
#define VQP_MIN_TILES 16          // C > 480: the three special intervals of a block need nst >= 8

template <int DT> struct ScreenPCfg {
    static constexpr int W = 8;                        // waves per workgroup
    static constexpr int NK = DT / 16;                 // MFMA k-steps per tile
    static constexpr int TILE_B = 64 * DT + 1024;      // fp16 tile: 32 codes x DT + 32 floats -||c||^2/2 (padded to 1 KiB)
    static constexpr int SUB = 2;                      // tiles per barrier interval
    static constexpr int SUPER_B = SUB * TILE_B;
    static constexpr int NCHUNK = SUPER_B / 1024;
    static constexpr int PMAX = (NCHUNK + W - 1) / W;  // 1-KiB pieces per wave and interval
    static constexpr int BUF_B = PMAX * W * 1024;      // LDS bytes per buffer (every wave copies PMAX pieces, unconditionally)
    static constexpr int PPS = (PMAX + SUB - 1) / SUB; // pieces a wave copies during one tile
    static size_t smem_bytes(int n_tiles16) { return 2 * (size_t)BUF_B + 2 * (size_t)n_tiles16 * 32 * 4 + 256; }
};

__device__ __forceinline__ void vqp_barrier()
{
    // LDS-only barrier: the ds_writes of this wave have landed (lgkmcnt), global loads stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <class F, int... I>
__device__ __forceinline__ void vqp_unroll(F &f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}

template <int DT, int METRIC, bool HASQ>
__global__ void __launch_bounds__(512, 2) vq_screenp_kernel(const ScreenArgs a, const int nsb, long long *const trace)
{
#ifdef VQP_TRACE            // dev build: s_memtime stamps of the first VQP_TRACE_IV intervals of workgroups 0 .. 7 (tools/trace_screenp.py)
#define VQP_TRACE_IV 48
    int trace_gi = 0;
#define VQP_STAMP(k) do { if (trace && blockIdx.x < 8 && trace_gi < VQP_TRACE_IV && lane == 0) \
        trace[(((size_t)blockIdx.x * 8 + wave) * VQP_TRACE_IV + trace_gi) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define VQP_STAMP(k) do {} while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Cfg = ScreenPCfg<DT>;
    constexpr int NK = Cfg::NK, TILE_B = Cfg::TILE_B, SUB = Cfg::SUB, SUPER_B = Cfg::SUPER_B;
    constexpr int PMAX = Cfg::PMAX, BUF_B = Cfg::BUF_B;
#ifndef VQP_PF
#define VQP_PF 2
#endif
    constexpr int PF = VQP_PF;                         // A-fragment ring depth
    static_assert(SUB == 2, "an interval is two tiles: two independent accumulator chains");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int half = lane >> 5;
    const int G = gridDim.x;
    const int nt16 = a.n_tiles16;
    const int nt16_bytes = nt16 * ScreenPCfg<DT>::TILE_B + 8192;   // tiles + the tail pad the unconditional copies over-read
    const int nst = nt16 / SUB;
    const int CP = nt16 * 32;                          // codes incl. padding
    float *const nrm = (float *)(smem + 2 * BUF_B);   // [2][CP] start values, scaled for the current / next iteration
    unsigned *const xch = (unsigned *)(nrm + 2 * CP); // [2][8] largest finite ||x||^2 bits of every wave
    const char *const tiles = a.tiles16;
    // the fp16 tiles as a buffer resource: staging loads are  buffer_load_dwordx4 v, v_lane_offset, s[rsrc], s_tile_offset offen
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tiles, 0, nt16_bytes, 0x00020000);
    const int sc = (int)a.scalars[2];
    const float y2max = __uint_as_float(a.scalars[0]);
    const float rmax = __uint_as_float(a.scalars[1]);
    const float ymax = sqrtf(y2max) * 1.0001f;
    const unsigned lane16 = (unsigned)lane * 16u;
    const unsigned wc = (unsigned)wave * (unsigned)(PMAX * 1024) + lane16;   // ... inside the wave's contiguous PMAX-KiB share of a buffer

    // ---- small helpers ------------------------------------------------------------------------------------------------------
    auto pick_sx = [&](unsigned mx) {                  // as in vq_screen16_kernel: largest finite ||x||^2 (float bits) -> norm below 2^14
        const int e2 = (int)(mx >> 23) - 127;
        int SX = (mx == 0u) ? 0 : 14 - ((e2 >> 1) + 1);
        SX = SX > 120 - sc ? 120 - sc : SX;
        SX = SX < -120 - sc ? -120 - sc : SX;
        SX = SX > sc + 90 ? sc + 90 : SX;
        return SX > 126 ? 126 : (SX < -126 ? -126 : SX);
    };
    auto eps_of = [&](float xs, int SX) {              // certificate threshold in unscaled units (vq_screen.hip header; one operand set)
        const float u = 5.9604645e-8f;
        const float conv = 2.f * 5.9604645e-8f * sqrtf((float)DT) * ymax * __uint_as_float((unsigned)(127 - SX) << 23);
        const float xn = sqrtf(xs) * 1.0001f;
        const float xy = xn * ymax;
        const float nacc = (float)(DT + 1);
        if (METRIC == 0) return u * (10.f * (xs + y2max + 2.f * xy) + 2.f * DT * xy + 4.f * nacc * 1.001f * (xy + 0.5f * y2max))
                                + 2.f * xn * rmax + conv + 4e-8f;
        return 2.f * (u * (DT + 2.f * DT) * 1.001f * xy + xn * rmax) + conv + 1e-30f;
    };
    auto conv_word = [&](unsigned w, float S) -> unsigned {   // two bf16 -> two fp16, exact above 2^-14 (scaled), truncated below
        const float lo = __uint_as_float(w << 16) * S, hi = __uint_as_float(w & 0xffff0000u) * S;
        return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lo, hi));
    };
    auto finite_bits = [&](float v) -> unsigned {
        const unsigned bits = __float_as_uint(v);
        return (bits & 0x7f800000u) == 0x7f800000u ? 0u : (bits & 0x7fffffffu);
    };
    auto xor32 = [&](unsigned v) -> unsigned {         // the value of lane ^ 32 (v_permlane32_swap: no index register, no LDS)
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return half ? r[0] : r[1];
    };
    auto xor32f = [&](float v) -> float { return __uint_as_float(xor32(__float_as_uint(v))); };
    auto wg_max = [&](const unsigned *p) -> unsigned { // max of the 8 waves' entries (uniform LDS reads)
        const uint4 u0 = *(const uint4 *)p, u1 = *(const uint4 *)(p + 4);
        const unsigned m0 = max(max(u0.x, u0.y), max(u0.z, u0.w)), m1_ = max(max(u1.x, u1.y), max(u1.z, u1.w));
        return (unsigned)__builtin_amdgcn_readfirstlane((int)max(m0, m1_));
    };
    auto rank_in = [&](unsigned long long mask) -> int {   // set bits of `mask` below this lane (v_mbcnt: no per-lane mask register)
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    };
    auto write_lists = [&](unsigned long long balo, unsigned long long balp, int base_o, int base_p, int cls, int row, int code, int id2)
        __attribute__((always_inline)) {
        // open rows fill the list from the front, pair rows (two candidate codes in the key) from the back, as in vq_screen16_kernel
        if (balo) {
            const int bo = __builtin_amdgcn_readfirstlane(base_o);
            if (cls == 1) {
                const int slot = bo + rank_in(balo);
                a.flag_rows[slot] = row;
                a.flag_keys[slot] = ~0ull;
            }
        }
        if (balp) {
            const int bp = __builtin_amdgcn_readfirstlane(base_p);
            if (cls == 2) {
                const int64_t slot = a.N - 1 - (bp + rank_in(balp));
                a.flag_rows[slot] = row;
                a.flag_keys[slot] = (unsigned long long)(unsigned)code | ((unsigned long long)(unsigned)id2 << 32);
            }
        }
    };
    auto scale_norms = [&](float *dst, float SS) {     // start values of one iteration: -||c||^2/2 * 2^(SX+sc), clamped to a finite -3e38
#pragma clang loop vectorize(disable) unroll(disable)
        for (int i = tid; i < CP; i += 512) {
            float v = *(const float *)(tiles + (size_t)(i >> 5) * TILE_B + 64 * DT + (i & 31) * 4);
            if (METRIC != 0) v = v < -1e38f ? v : 0.f;
            dst[i] = fmaxf(v * SS, -3.0e38f);
        }
    };

    // ---- registers that live across the whole kernel ---------------------------------------------------------------------------
    uint4 cur[NK], nxt[NK];                            // B operands of the block being swept / raw -> converted rows of the next one
    f32x16 accA, accB;
    // top 3 in TWO independent chains per lane (chain t sees the tiles 2 I + t): one chain alone is bound by the VALU's dependent-issue
    // latency (every v_med3 waits for the previous score's), two interleaved chains are not; merged at the end of the block
    float m1[2], m2[2], m3[2];
    int tix[2], tix2[2];

    // state of the block being swept
    int64_t row_c;
    bool rowok_c;
    float eps_c, SS_c, iSS_c;
    // state of the next block (built by the fillers)
    float xs2_n = 0.f;
    int SX_n = 0;
    // outputs of the previous block still to be written (q rows, list entries)
    bool has_prev = false;
    int pcode = 0, pcls = 0, pid2 = 0, prow = 0, pbase_o = 0, pbase_p = 0;
    unsigned long long pbalo = 0ull, pbalp = 0ull;
    int64_t prow0 = 0;
    int plim = 0;                                      // last valid row of the previous block, relative to prow0 (clamped to 31)
    char *pqbase = (char *)a.q_out;                    // q_out + prow0 * ldq * 2 (wave-uniform)
    const int ldq2 = (int)(a.ldq * 2);
    const int ldx2 = (int)(a.ldx * 2);
    const unsigned j16 = (unsigned)j * 16u;

    // ---- prologue: first block's rows, first codebook buffer, scale, start values ------------------------------------------------
    int sb = blockIdx.x;
    {
        const int64_t r00 = (int64_t)sb * 256 + wave * 32 + j;
        const int64_t r0 = r00 < a.N ? r00 : a.N - 1;
        const unsigned short *p = (const unsigned short *)a.x + r0 * a.ldx + 8 * half;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) nxt[ks] = *(const uint4 *)(p + ks * 16);
#pragma unroll
        for (int k = 0; k < 2 * PMAX; ++k)              // intervals 0 and 1 -> buffers 0 and 1 (role B only stages from interval 2 on)
            *(f32x4 *)(smem + (k / PMAX) * BUF_B + wc + (k % PMAX) * 1024) =
                *(const f32x4 *)(tiles + (size_t)(k / PMAX) * SUPER_B + wc + (size_t)(k % PMAX) * 1024);
        float xs = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const unsigned w[4] = {nxt[ks].x, nxt[ks].y, nxt[ks].z, nxt[ks].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[q]), __builtin_bit_cast(bf16x2, w[q]), xs, false);
        }
        xs += xor32f(xs);
        xs2_n = xs * 1.001f;
        if (tid < 16) xch[tid] = 0u;
        vqp_barrier();
        atomicMax(&xch[wave], finite_bits(xs2_n));
        vqp_barrier();
        SX_n = pick_sx(wg_max(xch));
        const float S = __uint_as_float((unsigned)(SX_n + 127) << 23);
        scale_norms(nrm, __uint_as_float((unsigned)(SX_n + sc + 127) << 23));
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
            cur[ks] = make_uint4(conv_word(nxt[ks].x, S), conv_word(nxt[ks].y, S), conv_word(nxt[ks].z, S), conv_word(nxt[ks].w, S));
        row_c = (int64_t)sb * 256 + wave * 32 + j;
        rowok_c = row_c < a.N;
        SS_c = __uint_as_float((unsigned)(SX_n + sc + 127) << 23);
        iSS_c = __uint_as_float((unsigned)(127 - SX_n - sc) << 23);
        eps_c = eps_of(xs2_n, SX_n);
        vqp_barrier();                                  // buffer 0 and the start values are in LDS for every wave
    }

    // one score of each tile into its chain's (best, second, third): the key carries the register number in its 4 low mantissa bits;
    // ONE asm statement (as builtins hipcc sinks every v_med3 of a tile behind the tile's last MFMA, vq_screen.hip), the two chains
    // interleaved instruction by instruction
    auto fold2 = [&](int e) __attribute__((always_inline)) {
        const float ka = __uint_as_float((__float_as_uint(accA[e]) & 0xfffffff0u) | (unsigned)e);
        const float kb = __uint_as_float((__float_as_uint(accB[e]) & 0xfffffff0u) | (unsigned)e);
        asm volatile("v_med3_f32 %2, %1, %2, %6\n\tv_med3_f32 %5, %4, %5, %7\n\t"
                     "v_med3_f32 %1, %0, %1, %6\n\tv_med3_f32 %4, %3, %4, %7\n\t"
                     "v_max_f32 %0, %0, %6\n\tv_max_f32 %3, %3, %7"
                     : "+v"(m1[0]), "+v"(m2[0]), "+v"(m3[0]), "+v"(m1[1]), "+v"(m2[1]), "+v"(m3[1]) : "v"(ka), "v"(kb));
    };
    auto book = [&](int t, float om1, float om2, int tile_id) __attribute__((always_inline)) {   // which tiles hold best / second of chain t
        const bool c1 = m1[t] != om1;
        const int from_old_best = (c1 && m2[t] == om1) ? tix[t] : tile_id;
        tix2[t] = (m2[t] != om2) ? from_old_best : tix2[t];
        tix[t] = c1 ? tile_id : tix[t];
        // pinned here: hipcc otherwise sinks every tile's bookkeeping to the end of the block and keeps a copy of (best, second)
        // per tile alive until then (64 registers at 32 tiles)
        asm volatile("" : "+v"(tix[t]), "+v"(tix2[t]));
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

    // Ping-pong roles: waves w and w + 4 share a SIMD.  Every wave runs the same stream  M(i) [X] F(i) [Y]  per interval i --
    // M = the interval's 32 MFMAs (two tiles = two independent accumulator chains, nothing but LDS reads between them: a VALU
    // instruction between two MFMAs of ONE chain costs ~43 cycles on gfx950), F = everything else (top-3 folds of the 32 scores,
    // staging, and the row / q fillers).  Role A (waves 0-3) takes its barrier at Y, role B (waves 4-7) at X: between two barriers
    // ("window" k) A runs M(k) F(k) and B runs F(k-1) M(k), so one wave's VALU phase faces its SIMD partner's MFMA phase.
    // LDS safety: M(k) of every wave falls into window k and reads buffer k & 1.  The tiles of interval k + 1 (buffer (k+1) & 1, last
    // read in window k - 1) are staged inside window k by everybody: role A in F(k), role B in F(k-1) -- i.e. role B's F(i) stages
    // interval i + 2 into the buffer its own M(i) has just finished with (every other wave's M(i) ended before the barrier at X).
#ifdef VQP_ONE_ROLE         // A/B: no ping-pong, every wave takes its barrier behind F
    const bool role_b = false;
#else
    const bool role_b = wave >= 4;
#endif
    const int stage_ahead = role_b ? 2 : 1;
    constexpr int BS0 = (PMAX + 1) / 2, BS1 = PMAX - BS0;   // staging pieces per wave and interval: carried over M / inside F
    f32x4 stgA[BS0];
#pragma unroll
    for (int i = 0; i < BS0; ++i)                    // the first F phase stages interval `stage_ahead`
        stgA[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trsrc, wc + i * 1024, (unsigned)stage_ahead * (unsigned)SUPER_B, 0));
    // the three special intervals of a block (the schedule is the same for every codebook size, nst >= 8)
    const int I_norm = nst >> 1;                       // ||x||^2 of the next block's rows (requested at interval 0) + scale exchange
    // ... and at I_norm + 2, I_norm + 3: scale agreed (>= one barrier later), start values + conversion of the rows in two halves

    for (int it = 0;; ++it, sb += G) {
        const bool has_next = sb + G < nsb;
        const float *const nrm_c = nrm + (it & 1) * CP + 4 * half;        // this iteration's start values (+ this half's 4 codes)
        float *const nrm_n = nrm + ((it + 1) & 1) * CP;
        unsigned *const xch_n = xch + ((it + 1) & 1) * 8;

#pragma unroll
        for (int t = 0; t < 2; ++t) { m1[t] = m2[t] = m3[t] = -__builtin_inff(); tix[t] = tix2[t] = 0; }

        // one barrier interval.  KIND selects the block-level work that rides in its F phase: 0 nothing, 1 request the next block's
        // rows, 2 list entries of the previous block, 3 row norms + scale exchange, 4 / 5 agree the scale + convert the rows.
        // (Separate instantiations instead of run-time branches inside one loop: a branch that redefines the 64 `nxt` registers
        //  makes hipcc shuffle all of them through copies at the join of EVERY interval.)
        auto interval = [&](const int I, auto Kc) __attribute__((always_inline)) {
            constexpr int KIND = decltype(Kc)::value;
            // ---- M: two tiles, 2 x 16 MFMAs, alternating accumulators ----
            VQP_STAMP(0);
            {
                const char *sbase = smem + (I & 1) * BUF_B;
                const float *nh = nrm_c + I * 64;
                const uint4 *ap0 = (const uint4 *)(sbase + lane16);
                const uint4 *ap1 = (const uint4 *)(sbase + TILE_B + lane16);
                uint4 a0[PF], a1[PF];
#pragma unroll
                for (int p = 0; p < PF; ++p) { a0[p] = ap0[p * 64]; a1[p] = ap1[p * 64]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {                              // start values -||c||^2/2 (scaled): the MFMAs' C operand
                    const f32x4 v = *(const f32x4 *)(nh + 8 * q), w = *(const f32x4 *)(nh + 32 + 8 * q);
                    accA[4 * q + 0] = v.x; accA[4 * q + 1] = v.y; accA[4 * q + 2] = v.z; accA[4 * q + 3] = v.w;
                    accB[4 * q + 0] = w.x; accB[4 * q + 1] = w.y; accB[4 * q + 2] = w.z; accB[4 * q + 3] = w.w;
                }
#pragma unroll
                for (int s_ = 0; s_ < NK; ++s_) {
#ifdef VQP_NO_MFMA      // A/B: the sweep without its matrix instructions
                    if (s_ > 0) continue;
#endif
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0[s_ % PF]), __builtin_bit_cast(f16x8, cur[s_]), accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1[s_ % PF]), __builtin_bit_cast(f16x8, cur[s_]), accB, 0, 0, 0);
                    if (s_ + PF < NK) { a0[s_ % PF] = ap0[(s_ + PF) * 64]; a1[s_ % PF] = ap1[(s_ + PF) * 64]; }
                    __builtin_amdgcn_sched_barrier(0);  // keeps the prefetch distance (hipcc otherwise sinks the ds_reads next to their use)
                }
            }
            VQP_STAMP(1);
            if (role_b) vqp_barrier();
            VQP_STAMP(2);

            // ---- F: staging, folds, fillers ----
            {
                int siv = I + stage_ahead;
                siv = siv >= nst ? siv - nst : siv;
                const unsigned gsrc = (unsigned)siv * (unsigned)SUPER_B;   // byte offset of the staged interval's tiles (SGPR: soffset)
                char *const ldst = smem + (siv & 1) * BUF_B + wc;
                // Staging, software-pipelined so that no wait ever faces an L2 round trip: the first BS0 pieces were requested at the
                // END of the previous F phase (a whole M phase ago) and go to LDS now; the other BS1 are requested now and go to LDS at
                // the end of this phase.  vmcnt retires in order, so the order of the requests matters: staging and the q gather (L2)
                // first, the next block's rows (HBM, interval 0 only) LAST -- nothing waits for them before the end of the NEXT F phase.
#ifndef VQP_NO_STAGE
#pragma unroll
                for (int i = 0; i < BS0; ++i) *(f32x4 *)(ldst + i * 1024) = stgA[i];
#endif
                f32x4 stgB[BS1];
#pragma unroll
                for (int i = 0; i < BS1; ++i) {
#ifdef VQP_NO_STAGE     // A/B: no L2 -> LDS traffic (the buffers keep the prologue's tiles)
                    stgB[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
                    stgB[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trsrc, wc + (BS0 + i) * 1024, gsrc, 0));
#endif
                }
                [[maybe_unused]] uint4 gq;
                const bool q_iv = HASQ && I < 16;
                if (q_iv) {                                  // q rows of the previous block: rows 2 I and 2 I + 1, one half-wave each
                    const int c0 = __builtin_amdgcn_ds_bpermute((2 * I + half) * 4, pcode);
                    gq = *(const uint4 *)((const char *)a.embed_bf16 + ((unsigned)c0 * (unsigned)(DT * 2) + j16));
                }
                VQP_STAMP(5);
                // folds of the two tiles, one chain each
                {
                    const float o10 = m1[0], o20 = m2[0], o11 = m1[1], o21 = m2[1];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
#ifndef VQP_NO_FOLD      // A/B
                        fold2(e);
#else
                        if (e == 0) fold2(e);
#endif
                    }
                    book(0, o10, o20, 2 * I);
                    book(1, o11, o21, 2 * I + 1);
                }
                VQP_STAMP(6);
#ifndef VQP_NO_STAGE
#pragma unroll
                for (int i = 0; i < BS1; ++i) *(f32x4 *)(ldst + (BS0 + i) * 1024) = stgB[i];
#endif
                if (q_iv && has_prev) {
                    // rows past the end repeat the last row (same code, same bytes): row offset clamped to plim = N - 1 - prow0
                    const int r0 = min(2 * I + half, plim);
                    *(uint4 *)(pqbase + ((unsigned)(r0 * ldq2) + j16)) = gq;
                }
                {   // the first pieces of the NEXT F phase's interval
                    int siv1 = siv + 1;
                    siv1 = siv1 >= nst ? siv1 - nst : siv1;
                    const unsigned gsrc1 = (unsigned)siv1 * (unsigned)SUPER_B;
#ifndef VQP_NO_STAGE
#pragma unroll
                    for (int i = 0; i < BS0; ++i)
                        stgA[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trsrc, wc + i * 1024, gsrc1, 0));
#endif
                }
                VQP_STAMP(7);
                // ---- the special intervals ----
                if constexpr (KIND == 1) {
                    if (lane == 0) xch_n[wave] = 0u;         // last read two iterations ago
                    // the next block's rows: wave-uniform base (SGPRs) + 32-bit lane offset; rows past the end repeat the last valid one
                    const int64_t rb = (int64_t)(sb + G < nsb ? sb + G : nsb - 1) * 256 + wave * 32;
                    const int64_t lim = a.N - 1 - rb;
                    const char *const xb = (const char *)a.x + (rb < a.N ? rb : a.N - 1) * a.ldx * 2;
                    unsigned xo = (unsigned)min(j, lim > 31 ? 31 : (lim < 0 ? 0 : (int)lim)) * (unsigned)ldx2 + (unsigned)half * 16u;
                    asm volatile("" : "+v"(xo));             // formed here (hoisted out of the interval loop, the 16 addresses cost 32 VGPRs)
                    const char *const xl = xb + xo;          // SGPR base + 32-bit lane offset; the k-step is the instruction's immediate
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) nxt[ks] = *(const uint4 *)(xl + ks * 32);
                } else if constexpr (KIND == 2) {
                    if (has_prev) write_lists(pbalo, pbalp, pbase_o, pbase_p, pcls, prow, pcode, pid2);   // (its atomics have returned)
                } else if constexpr (KIND == 3) {
                    float xs = 0.f;
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) {
                        const unsigned w[4] = {nxt[ks].x, nxt[ks].y, nxt[ks].z, nxt[ks].w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            xs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[q]), __builtin_bit_cast(bf16x2, w[q]), xs, false);
                    }
                    xs += xor32f(xs);
                    xs2_n = xs * 1.001f;
                    atomicMax(&xch_n[wave], finite_bits(xs2_n));
                } else if constexpr (KIND == 4 || KIND == 5) {
                    if constexpr (KIND == 4) {
                        SX_n = pick_sx(wg_max(xch_n));
                        scale_norms(nrm_n, __uint_as_float((unsigned)(SX_n + sc + 127) << 23));
                    }
                    const float Sn = __uint_as_float((unsigned)(SX_n + 127) << 23);
                    // bf16 -> scaled fp16 in place, one half of the registers per interval.  (The empty asm pins the conversion HERE:
                    // its only consumer is the copy at the end of the block, and hipcc would sink all the conversions down there.)
                    if constexpr (KIND == 4) {
#pragma unroll
                        for (int ks = 0; ks < NK / 2; ++ks) {
                            nxt[ks] = make_uint4(conv_word(nxt[ks].x, Sn), conv_word(nxt[ks].y, Sn), conv_word(nxt[ks].z, Sn), conv_word(nxt[ks].w, Sn));
                            asm volatile("" : "+v"(nxt[ks].x), "+v"(nxt[ks].y), "+v"(nxt[ks].z), "+v"(nxt[ks].w));
                        }
                    } else {
#pragma unroll
                        for (int ks = NK / 2; ks < NK; ++ks) {
                            nxt[ks] = make_uint4(conv_word(nxt[ks].x, Sn), conv_word(nxt[ks].y, Sn), conv_word(nxt[ks].z, Sn), conv_word(nxt[ks].w, Sn));
                            asm volatile("" : "+v"(nxt[ks].x), "+v"(nxt[ks].y), "+v"(nxt[ks].z), "+v"(nxt[ks].w));
                        }
                    }
                }
            }
            VQP_STAMP(3);
            if (!role_b) vqp_barrier();
            VQP_STAMP(4);
#ifdef VQP_TRACE
            ++trace_gi;
#endif
        };
        using K0 = std::integral_constant<int, 0>;
        interval(0, std::integral_constant<int, 1>{});
        interval(1, K0{});
        interval(2, std::integral_constant<int, 2>{});
#pragma unroll 1
        for (int I = 3; I < I_norm; ++I) interval(I, K0{});
        interval(I_norm, std::integral_constant<int, 3>{});
        interval(I_norm + 1, K0{});
        interval(I_norm + 2, std::integral_constant<int, 4>{});
        interval(I_norm + 3, std::integral_constant<int, 5>{});
#pragma unroll 1
        for (int I = I_norm + 4; I < nst; ++I) interval(I, K0{});

        // ---- merge the half-waves, classify (certified / pair / open), emit the index ----
        int code, cls, id2;
        {
            int ic[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int e1 = (int)(__float_as_uint(m1[t]) & 15u), e2b = (int)(__float_as_uint(m2[t]) & 15u);
                ic[t][0] = tix[t] * 32 + 8 * (e1 >> 2) + 4 * half + (e1 & 3);
                ic[t][1] = tix2[t] * 32 + 8 * (e2b >> 2) + 4 * half + (e2b & 3);
            }
            float a1, a2, a3, b1, b2, b3;
            int ia1, ia2;
            merge3(m1[0], m2[0], m3[0], ic[0][0], ic[0][1], m1[1], m2[1], m3[1], ic[1][0], ic[1][1], a1, a2, a3, ia1, ia2);   // the two chains
            merge3(a1, a2, a3, ia1, ia2, xor32f(a1), xor32f(a2), xor32f(a3), (int)xor32((unsigned)ia1), (int)xor32((unsigned)ia2),
                   b1, b2, b3, code, id2);                                                                                  // the two half-waves
            const float thr = eps_c * SS_c + 8e-6f * fabsf(b1);
            const bool certified = ((b1 - b2) > thr) && code < a.C;
            const bool pair = !certified && ((b1 - b3) > thr) && code < a.C && id2 < a.C;
            if (rowok_c && half == 0) {
                a.idx_out[row_c * a.idx_stride] = (int64_t)(code < a.C ? code : 0);
                if (a.dbg) {
                    float *d = a.dbg + row_c * 4;
                    d[0] = b1 * iSS_c; d[1] = b2 * iSS_c; d[2] = thr * iSS_c; d[3] = certified ? 0.f : (pair ? 2.f : 1.f);
                }
            }
            if (code >= a.C) code = 0;
            const bool on = rowok_c && half == 0;
            cls = !on ? 0 : (certified ? 0 : (pair ? 2 : 1));
        }
        pbalo = __ballot(cls == 1);
        pbalp = __ballot(cls == 2);
        pbase_o = 0; pbase_p = 0;
        if (lane == 0) {
            if (pbalo) pbase_o = atomicAdd(a.flag_count, (int)__popcll(pbalo));
            if (pbalp) pbase_p = atomicAdd(a.flag_count + 1, (int)__popcll(pbalp));
        }
        pcode = code; pcls = cls; pid2 = id2; prow = (int)row_c;
        prow0 = (int64_t)sb * 256 + wave * 32;
        {
            const int64_t lim = a.N - 1 - prow0;
            plim = lim > 31 ? 31 : (lim < 0 ? 0 : (int)lim);
            pqbase = (char *)a.q_out + (prow0 < a.N ? prow0 : a.N - 1) * a.ldq * 2;
        }
        has_prev = true;
        if (!has_next) break;

        // ---- the next block becomes the current one ----
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) cur[ks] = nxt[ks];
        row_c = (int64_t)(sb + G) * 256 + wave * 32 + j;
        rowok_c = row_c < a.N;
        SS_c = __uint_as_float((unsigned)(SX_n + sc + 127) << 23);
        iSS_c = __uint_as_float((unsigned)(127 - SX_n - sc) << 23);
        eps_c = eps_of(xs2_n, SX_n);
    }

    // ---- tail: the last block's q rows and list entries ----
    if (HASQ) {
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
            const int c = __builtin_amdgcn_ds_bpermute((2 * t + half) * 4, pcode);
            const uint4 g = *(const uint4 *)((const char *)a.embed_bf16 + ((unsigned)c * (unsigned)(DT * 2) + j16));
            const int rl = min(2 * t + half, plim);
            *(uint4 *)(pqbase + ((unsigned)(rl * ldq2) + j16)) = g;
        }
    }
    {
        write_lists(pbalo, pbalp, pbase_o, pbase_p, pcls, prow, pcode, pid2);
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
#ifndef VQP_MIN_ROWS
#define VQP_MIN_ROWS (256 * 256)      // below one super-block per CU the 4-wave kernel spreads the rows better
#endif

static long long *vqp_g_trace = nullptr;              // dev builds with -DVQP_TRACE: where the kernel puts its s_memtime stamps
extern "C" void vqhip_screenp_set_trace(long long *p) { vqp_g_trace = p; }

// Opt-in (VQHIP_SCREEN_PERSIST=1).  Measured on MI355X at cfg 2 (round 3): bit-identical to vq_screen16_kernel
// on every case of tools/persist_check.py, but 621 / 682 us (index / index + q) against 554 / 619 us -- the same VALU issue time
// (SQ_ACTIVE_INST_VALU 127 M vs 128 M quad-cycles), 1.9x the parked-wave time (SQ_WAIT_ANY 266 M vs 141 M): with ONE row block per
// wave every A fragment feeds one MFMA instead of two, and a wave's M and F phases add up serially whatever its partner does.
static int vqp_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("VQHIP_SCREEN_PERSIST"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

int vq_screenp_eligible(const ScreenArgs &a, int x_dtype, int DT)
{
    if (!vqp_enabled()) return 0;
    if (x_dtype != VQHIP_BF16 || DT != 256) return 0;
    if (a.n_tiles16 < VQP_MIN_TILES || a.n_tiles16 * 32 > VQP_MAX_NORM_CODES) return 0;
    if (a.resid_out || a.sqerr_partial || a.prev_idx) return 0;
    if (a.N < VQP_MIN_ROWS) return 0;
    if (a.q_out && ((((uintptr_t)a.q_out) & 15) || ((a.ldq * 2) & 15))) return 0;
    return 1;
}

template <int METRIC, bool HASQ>
static int vqp_launch(const ScreenArgs &a, hipStream_t st)
{
    using Cfg = ScreenPCfg<256>;
    static VqAttrOnce once;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) n_cu = 256;
        else n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const size_t smem = Cfg::smem_bytes(a.n_tiles16);
    if (int rc = vq_set_max_smem(once, (const void *)vq_screenp_kernel<256, METRIC, HASQ>, 160 * 1024, "vq_screenp_kernel")) return rc;
    const int nsb = (int)((a.N + 255) / 256);
    const int grid = nsb < n_cu ? nsb : n_cu;
    hipLaunchKernelGGL((vq_screenp_kernel<256, METRIC, HASQ>), dim3((unsigned)grid), dim3(512), smem, st, a, nsb, vqp_g_trace);
    return vq_launch_status("vq_screenp_kernel");
}

int vq_screenp_launch(const ScreenArgs &a, int metric_is_cosine, hipStream_t st)
{
    if (a.q_out) return metric_is_cosine ? vqp_launch<1, true>(a, st) : vqp_launch<0, true>(a, st);
    return metric_is_cosine ? vqp_launch<1, false>(a, st) : vqp_launch<0, false>(a, st);
}
