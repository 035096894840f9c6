// vq_screen.hip -- screened nearest-code assignment, gfx950 only.  D in {32, 64, 128, 256}; bf16 rows (vq_screen_kernel)
// and fp32 rows (vq_screen_f32_kernel, further down); Euclidean metric, or cosine on unit-norm rows (METRIC 1).
//
// The exact kernel (vqhip.hip, vq_assign_kernel) evaluates the reference's cdist (vqp.py:58-62) bit for bit on the
// fp32 MFMA pipe, which runs at 1/16 of the bf16 MFMA rate.  The same INDICES can be had much cheaper, still exactly
// (described for bf16 rows and the Euclidean metric; the other variants state their differences where they are defined):
//
//   1. screen (this file): bf16 rows are exact MFMA operands; the fp32 codebook is split c = c_hi + c_lo into two
//      bf16 parts (bf16 keeps 8 significant bits: |c - c_hi - c_lo| <= 2^-16 |c|) and  t[n, c] = x_n . c_hi + x_n . c_lo - ||c||^2 / 2  is
//      accumulated by v_mfma_f32_32x32x16_bf16 (2 passes at 16x the fp32 rate).  argmax_c t = argmin_c cdist up to an
//      error eps(n) that is bounded below; per row the kernel keeps the best and the second best t.
//   2. a row whose margin (best - second) exceeds the bound has a certified winner: every other code is farther in
//      the reference's own fp32 arithmetic as well, ties of the rounded sqrt included.  Its index, its gathered code
//      and its squared error are final.
//   3. the few rows that are not certified (a fraction of a percent to a few percent) are appended to a list and
//      re-done by the exact kernel (vq_assign_listed), which overwrites their outputs.
//
// Error bound, in units of s = ||x||^2 + ||c||^2 - 2 x.c (u = 2^-24, D features, X = ||x||, Y = max_c ||c||):
//   reference chain (oracle/vq_oracle.c::vqo_assign):  |s_ref - s| <= u (x2 + y2) + u s + 2 D u X Y
//   screen:  split 2 * 2^-16 X Y = 512 u X Y;  accumulation of 2 D products + the initial value inside the MFMAs, modelled as one
//            rounding per added term with TRUNCATION (2u) -- pessimistic for a fused dot-product unit --
//            2 * 2 D * 2u * (X Y + Y^2 / 2) = 8 D u (X Y + Y^2 / 2)
//   sqrt collapse: distances that differ by < 4 ulp(s) may round to the same sqrt (vqp.py:62) and tie: 8 u s
//   index bits: the kernel stores the lane-local code number in the 4 low mantissa bits of t: 2 * 16 ulp(t)
// A row is certified when  t_best - t_second > eps_t,  eps_t = eps_s (an s margin of 2 eps_s).  tests/test_gpu_ops.py
// measures the actual |t - t_exact| against this bound (observed: below 5 % of it) and checks indices bit for bit.

#include <math.h>
#include <string.h>

#include "vqhip_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef VQS_PF
#define VQS_PF 4             // depth of the A-fragment ring (LDS -> VGPR prefetch distance in steps of 2 MFMAs)
#define VQS_PIN 1
#endif
#ifndef VQS_OUT1
#define VQS_OUT2 1           // output phase ordering (see below); -DVQS_OUT1 selects the first version for A/B runs
#endif
#ifndef VQS_WAVES
#define VQS_WAVES 4          // waves per workgroup (2 workgroups of 4 or 1 of 8 per CU: 2 waves per SIMD either way)
#endif

#define VQ_SCREEN_ROWS (VQS_WAVES * 64)   // rows per workgroup: waves x 2 row blocks x 32

struct ScreenArgs {
    const void *x;                     // rows, bf16 (vq_screen_kernel) or fp32 (vq_screen_f32_kernel)
    int64_t N;
    int64_t ldx;
    const char *tiles;                 // screening tiles inside the packed codebook
    const unsigned short *embed_bf16;  // bf16 codebook copy inside the packed codebook
    const float *embed;                // fp32 codebook (q rows of fp32 I/O)
    const unsigned *scalars;           // [0] = float bits of max ||c||^2
    int C;
    int n_tiles;
    int64_t *idx_out;
    void *q_out;                       // nullable, x's dtype
    int64_t ldq;
    void *resid_out;                   // nullable, x's dtype: x - q (the next residual-VQ stage's input, rvq.py:524)
    int64_t ldr;
    double *sqerr_partial;             // nullable, one entry per workgroup
    const uint8_t *row_mask;
    int *flag_count;
    int *flag_rows;
    unsigned long long *flag_keys;     // [N] keys of the exact pass, preset to ~0 for every appended row
    float *dbg;                        // nullable [N, 4]: t_best, t_second, eps_t, flagged
#ifdef VQ_TRACE
    long long *trace;
#endif
};

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

// best / second best of a 16-score accumulator.  key = score with its 4 low mantissa bits replaced by the register
// number, so one v_med3 + one v_max per score track both values AND the position of the best.
__device__ __forceinline__ void top2_tile(const f32x16 &acc, float &m1, float &m2, int &tix, int ct)
{
    const float om = m1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float k = __uint_as_float((__float_as_uint(acc[e]) & 0xfffffff0u) | (unsigned)e);
        m2 = __builtin_amdgcn_fmed3f(m1, m2, k);      // m1 >= m2: the median is the new runner-up
        // fmaxf() on a value made by integer ops costs an extra canonicalising v_max_f32 k, k, k: emit the bare instruction
        asm("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(m1), "v"(k));
    }
    tix = (m1 != om) ? ct : tix;
}

// METRIC 0: Euclidean (t = x.c - ||c||^2 / 2).  METRIC 1: cosine on rows that are already unit-norm (t = x.c, the
// reference's einsum at vqp.py:741; no sqrt, ties only between equal floats): the accumulator starts at 0 and the bound has
// no norm terms -- reference chain D u XY, split 2^-16 XY, accumulation 4 D u XY, and the margin must cover both codes: 2x.
template <int DT, int METRIC>
__global__ void __launch_bounds__(VQS_WAVES * 64, 8 / VQS_WAVES) vq_screen_kernel(const ScreenArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_B = 128 * DT + 1024;
    constexpr int NCHUNK = TILE_B / 1024;          // 1-KiB pieces per tile
    constexpr int NK = DT / 16;                    // MFMA k-steps
    constexpr int STEPS = 2 * NK;                  // (k-step, hi/lo part) pairs, 2 MFMAs each
    constexpr int PMAX = (NCHUNK + VQS_WAVES - 1) / VQS_WAVES;   // pieces per wave
#ifndef VQS_NB
#define VQS_NB 2
#endif
    constexpr int NB = (STEPS >= 2 * VQS_NB) ? VQS_NB : 2;   // staging batches
    constexpr int BS = (PMAX + NB - 1) / NB;
    constexpr int HALF = STEPS / NB;               // steps between batch starts
#ifdef VQS_LAGM
    constexpr int LAG = (HALF > VQS_LAGM + 1) ? HALF - VQS_LAGM : 1;
#else
    constexpr int LAG = (HALF > 3) ? HALF - 2 : 1; // steps between a batch's loads and its LDS stores
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int half = lane >> 5;
    const int64_t wrow0 = (int64_t)blockIdx.x * VQ_SCREEN_ROWS + wave * 64;
    VQ_PHASE(0);

    // ---- tile 0: wave w copies the 1-KiB pieces w, w + WAVES, ... ----
    constexpr int PSTRIDE = VQS_WAVES * 1024;
    const int my_pieces = (NCHUNK - wave + VQS_WAVES - 1) / VQS_WAVES;
    const int piece_off = wave * 1024 + lane * 16;
    for (int k = 0; k < my_pieces; ++k)
        *(f32x4 *)(smem + piece_off + k * PSTRIDE) = *(const f32x4 *)(a.tiles + piece_off + (size_t)k * PSTRIDE);

    // ---- x rows -> B operands: lane (j, half) holds x[row][16 ks + 8 half + 0..7] for every k-step ----
    uint4 xb[2][NK];
    int64_t rows[2];
    bool row_ok[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        rows[rb] = wrow0 + rb * 32 + j;
        row_ok[rb] = rows[rb] < a.N;
        const int64_t rc = row_ok[rb] ? rows[rb] : (a.N - 1);
        const unsigned short *p = (const unsigned short *)a.x + rc * a.ldx + 8 * half;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) xb[rb][ks] = *(const uint4 *)(p + ks * 16);
    }

    // ---- ||x||^2 (any order: it only scales the error bound) ----
    float eps[2];
    {
        const float y2max = __uint_as_float(a.scalars[0]);
        const float ymax = sqrtf(y2max) * 1.0001f;
        const float u = 5.9604645e-8f;   // 2^-24
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
            xs *= 1.001f;
            const float xn = sqrtf(xs) * 1.0001f;
            const float xy = xn * ymax;
            // eps_s (see the header): 10 u (x2 + y2max + 2 xy)  >=  u (x2 + y2) + 9 u s
            if (METRIC == 0) eps[rb] = u * (10.f * (xs + y2max + 2.f * xy) + (2.f * DT + 512.f) * xy + 8.f * DT * (xy + 0.5f * y2max)) + 4e-8f;
            else             eps[rb] = 2.f * u * (5.f * DT + 256.f) * xy + 1e-30f;
        }
    }

    float m1[2] = {-__builtin_inff(), -__builtin_inff()};
    float m2[2] = {-__builtin_inff(), -__builtin_inff()};
    int tix[2] = {0, 0};
    VQ_PHASE(1);   // x loaded, eps computed

    const int nt = a.n_tiles;
    for (int ct = 0; ct < nt; ++ct) {
        const int buf = ct & 1;
        VQ_STAMP(0);
        __syncthreads();   // tile ct has landed for every wave; the other buffer is free
        VQ_STAMP(1);
        const char *tile = smem + buf * TILE_B;
        const bool more = ct + 1 < nt;
        const char *gsrc = a.tiles + (size_t)(more ? ct + 1 : ct) * TILE_B + piece_off;
        char *ldst = smem + (buf ^ 1) * TILE_B + piece_off;
        const int npieces = more ? my_pieces : 0;

        // accumulators start at -||c||^2 / 2 of the register's code (same mapping as the fp32 kernel:
        // register e <-> code 8 (e >> 2) + 4 half + (e & 3) of the tile)
        const float *nh = (const float *)(tile + 128 * DT);
        f32x16 acc0, acc1;
        if (METRIC == 0 || (ct == nt - 1 && (a.C & 31))) {   // cosine starts at 0; only a ragged last tile needs the
#pragma unroll                                               // -3e38 of its padding codes (they would score 0 otherwise)
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *(const f32x4 *)(nh + 8 * q + 4 * half);
                if (METRIC != 0) { v.x = v.x < -1e38f ? v.x : 0.f; v.y = v.y < -1e38f ? v.y : 0.f;
                                   v.z = v.z < -1e38f ? v.z : 0.f; v.w = v.w < -1e38f ? v.w : 0.f; }
                acc0[4 * q + 0] = v.x; acc0[4 * q + 1] = v.y; acc0[4 * q + 2] = v.z; acc0[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
        }
        acc1 = acc0;

        const uint4 *ap = (const uint4 *)tile + lane;
        f32x4 stg[BS];
        uint4 af[VQS_PF];   // A-fragment ring: the ds_read of step s + VQS_PF is issued behind the MFMAs of step s
#pragma unroll
        for (int p = 0; p < VQS_PF; ++p) af[p] = ap[(p < STEPS ? p : 0) * 64];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int ks = s >> 1;
            const bf16x8 av = __builtin_bit_cast(bf16x8, af[s % VQS_PF]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, xb[0][ks]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, xb[1][ks]), acc1, 0, 0, 0);
            if (s + VQS_PF < STEPS) af[s % VQS_PF] = ap[(s + VQS_PF) * 64];
#ifdef VQS_PIN
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distance: hipcc otherwise sinks the ds_reads next to their use
#endif
#ifdef VQS_GLDS
            // LDS-DMA variant: piece s of this wave's share goes L2 -> LDS directly (no staging registers, no ds_write);
            // the wave waits for its DMAs at the end of the tile, before the barrier that publishes the buffer
            if (s < PMAX && s < npieces)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)s * PSTRIDE),
                                                 (__attribute__((address_space(3))) void *)(ldst - lane * 16 + s * PSTRIDE), 16, 0, 0);
#elif !defined(VQS_NO_STAGE)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (s == b * HALF) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < BS; ++i)   // unconditional: a piece past this wave's share reads the tail pad
                        if (b * BS + i < PMAX) stg[i] = *(const f32x4 *)(gsrc + (size_t)(b * BS + i) * PSTRIDE);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s == b * HALF + LAG) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < BS; ++i)
                        if (b * BS + i < npieces) *(f32x4 *)(ldst + (b * BS + i) * PSTRIDE) = stg[i];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#endif
        }
#ifdef VQS_GLDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        VQ_STAMP(2);
#ifndef VQS_NO_EPI
        top2_tile(acc0, m1[0], m2[0], tix[0], ct);
        top2_tile(acc1, m1[1], m2[1], tix[1], ct);
        VQ_STAMP(3);
#else
        m1[0] = fmaxf(m1[0], acc0[ct & 15]); m1[1] = fmaxf(m1[1], acc1[ct & 15]);
#endif
    }

    VQ_PHASE(2);   // sweep done
    // ---- merge the half-waves, certify, emit ----
    int code[2];
    bool flagged[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int e = (int)(__float_as_uint(m1[rb]) & 15u);
        const int c_own = tix[rb] * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
        const float o1 = __shfl_xor(m1[rb], 32, 64);
        const float o2 = __shfl_xor(m2[rb], 32, 64);
        const int oc = __shfl_xor(c_own, 32, 64);
        const bool take = o1 > m1[rb];
        const float b1 = take ? o1 : m1[rb];
        const float lo1 = take ? m1[rb] : o1;
        const float b2 = fmaxf(lo1, fmaxf(m2[rb], o2));
        code[rb] = take ? oc : c_own;
        const float thr = eps[rb] + 8e-6f * fabsf(b1);
        flagged[rb] = !((b1 - b2) > thr) || code[rb] >= a.C;
        if (row_ok[rb] && half == 0) {
            a.idx_out[rows[rb]] = (int64_t)(code[rb] < a.C ? code[rb] : 0);
            if (a.dbg) {
                float *d = a.dbg + rows[rb] * 4;
                d[0] = b1; d[1] = b2; d[2] = thr; d[3] = flagged[rb] ? 1.f : 0.f;
            }
        }
        if (code[rb] >= a.C) code[rb] = 0;
        // append the uncertified rows to the list (one atomic per wave and row block)
        const bool f = flagged[rb] && row_ok[rb] && half == 0;
        const unsigned long long bal = __ballot(f);
        if (bal) {
            int base = 0;
            if (lane == 0) base = atomicAdd(a.flag_count, (int)__popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (f) {
                const int slot = base + (int)__popcll(bal & ((1ull << lane) - 1ull));
                a.flag_rows[slot] = (int)rows[rb];
                a.flag_keys[slot] = ~0ull;
            }
        }
    }

    VQ_PHASE(3);   // idx + list written
    // ---- outputs per row block: the loss operands (this lane's slice of its row's code, L2) are requested first so
    //      their latency hides behind the q copy; q = bf16 codebook rows written as whole rows, 16 rows in flight;
    //      squared error of the certified rows from the registers (the listed rows are counted by the exact pass) ----
    double ds = 0.0;
#ifdef VQS_OUT2
    if (!a.resid_out) {
        // loss(rb 0) -> [q rows of rb 0, 32 in flight || loss operands of rb 1] -> loss(rb 1) -> q rows of rb 1, 32 in flight:
        // three load round trips, no batch waits for the previous batch's stores (its registers are not reused)
        auto loss_of = [&](int rb, const uint4 (&gq)[NK]) {
            f32x2 ls = {0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const unsigned gw[4] = {gq[ks].x, gq[ks].y, gq[ks].z, gq[ks].w};
                const unsigned xw[4] = {xb[rb][ks].x, xb[rb][ks].y, xb[rb][ks].z, xb[rb][ks].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 gv = {__uint_as_float(gw[q] << 16), __uint_as_float(gw[q] & 0xffff0000u)};
                    const f32x2 xv = {__uint_as_float(xw[q] << 16), __uint_as_float(xw[q] & 0xffff0000u)};
                    const f32x2 df = gv - xv;
                    ls = __builtin_elementwise_fma(df, df, ls);
                }
            }
            const bool counted = row_ok[rb] && !flagged[rb] && (!a.row_mask || a.row_mask[rows[rb]] != 0);
            ds += counted ? (double)(ls[0] + ls[1]) : 0.0;
        };
        uint4 gq0[NK], gq1[NK];
        if (a.sqerr_partial) {
            const unsigned short *er = a.embed_bf16 + (size_t)code[0] * DT + 8 * half;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) gq0[ks] = *(const uint4 *)(er + ks * 16);
            loss_of(0, gq0);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned short *er1 = a.embed_bf16 + (size_t)code[1] * DT + 8 * half;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) gq1[ks] = *(const uint4 *)(er1 + ks * 16);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            if (a.q_out) {
                uint2 g[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int c = __builtin_amdgcn_readlane(code[rb], u);
                    if (lane * 4 < DT) g[u] = *(const uint2 *)(a.embed_bf16 + (size_t)c * DT + lane * 4);
                }
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int64_t rr = wrow0 + rb * 32 + u;
                    if (rr < a.N && lane * 4 < DT) *(uint2 *)((unsigned short *)a.q_out + rr * a.ldq + lane * 4) = g[u];
                }
            }
            if (rb == 0 && a.sqerr_partial) {
                loss_of(1, gq1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
#endif
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        uint4 gq[NK];
        if (a.sqerr_partial) {
            const unsigned short *er = a.embed_bf16 + (size_t)code[rb] * DT + 8 * half;
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) gq[ks] = *(const uint4 *)(er + ks * 16);
        }
        if (a.q_out || a.resid_out) {
            const bool want_r = a.resid_out != nullptr;
#pragma unroll
            for (int r0 = 0; r0 < 32; r0 += 16) {
                uint2 g[16], xv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int c = __builtin_amdgcn_readlane(code[rb], r0 + u);
                    const int64_t rr = wrow0 + rb * 32 + r0 + u;
                    if (lane * 4 < DT) {
                        g[u] = *(const uint2 *)(a.embed_bf16 + (size_t)c * DT + lane * 4);
                        if (want_r) xv[u] = *(const uint2 *)((const unsigned short *)a.x + (rr < a.N ? rr : a.N - 1) * a.ldx + lane * 4);
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int64_t rr = wrow0 + rb * 32 + r0 + u;
                    if (rr < a.N && lane * 4 < DT) {
                        if (a.q_out) *(uint2 *)((unsigned short *)a.q_out + rr * a.ldq + lane * 4) = g[u];
                        if (want_r) *(uint2 *)((unsigned short *)a.resid_out + rr * a.ldr + lane * 4) = vq_bf16x4_sub(xv[u], g[u]);
                    }
                }
            }
        }
        if (a.sqerr_partial) {
            f32x2 ls = {0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const unsigned gw[4] = {gq[ks].x, gq[ks].y, gq[ks].z, gq[ks].w};
                const unsigned xw[4] = {xb[rb][ks].x, xb[rb][ks].y, xb[rb][ks].z, xb[rb][ks].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 gv = {__uint_as_float(gw[q] << 16), __uint_as_float(gw[q] & 0xffff0000u)};
                    const f32x2 xv = {__uint_as_float(xw[q] << 16), __uint_as_float(xw[q] & 0xffff0000u)};
                    const f32x2 df = gv - xv;
                    ls = __builtin_elementwise_fma(df, df, ls);
                }
            }
            const bool counted = row_ok[rb] && !flagged[rb] && (!a.row_mask || a.row_mask[rows[rb]] != 0);
            ds += counted ? (double)(ls[0] + ls[1]) : 0.0;
        }
    }
    VQ_PHASE(4);   // q rows + squared error done
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
// fp32 rows.  x is split as well, x = x_hi + x_mid + r_x (bf16 parts, |r_x| <= 2^-16 |x|), and three products are
// accumulated per k-step: c_hi x_hi, c_hi x_mid, c_lo x_hi.  Dropped: c_lo x_mid, c r_x, r_c x -- each <= 2^-16 |x||c|
// (+ second order), so the split term of the bound becomes 3.03 * 2 * 2^-16 X Y <= 1600 u X Y and the accumulation term
// 12 D u (3 D products).  One 32-row block per wave (the two operand sets of a row block fill the registers the bf16
// kernel spends on a second row block), VQS_F32_WAVES waves per workgroup; the three MFMAs of a k-step alternate between
// two accumulators so that no MFMA waits for its predecessor.  q (fp32 code rows) and the squared error are produced by
// a row-cooperative pass that re-reads x, because the registers only hold x to 16 bits.
// ------------------------------------------------------------------------------------------------
#ifndef VQS_F32_WAVES
#define VQS_F32_WAVES 8
#endif
#define VQ_SCREEN_F32_ROWS (VQS_F32_WAVES * 32)

__device__ __forceinline__ void split8_bf16(const f32x4 &v0, const f32x4 &v1, uint4 &h, uint4 &m)
{
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned hw[4], mw[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const unsigned short h0 = vq_f32_to_bf16_rne(v[e]), h1 = vq_f32_to_bf16_rne(v[e + 1]);
        const unsigned short m0 = vq_f32_to_bf16_rne(v[e] - vq_bf16_bits_to_f32(h0));       // v - h is exact in fp32
        const unsigned short m1 = vq_f32_to_bf16_rne(v[e + 1] - vq_bf16_bits_to_f32(h1));
        hw[e >> 1] = (unsigned)h0 | ((unsigned)h1 << 16);
        mw[e >> 1] = (unsigned)m0 | ((unsigned)m1 << 16);
    }
    h = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    m = make_uint4(mw[0], mw[1], mw[2], mw[3]);
}

template <int DT, int METRIC>
__global__ void __launch_bounds__(VQS_F32_WAVES * 64, 8 / VQS_F32_WAVES) vq_screen_f32_kernel(const ScreenArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int W = VQS_F32_WAVES;
    constexpr int TILE_B = 128 * DT + 1024;
    constexpr int NCHUNK = TILE_B / 1024;
    constexpr int NK = DT / 16;
    constexpr int STEPS = 2 * NK;
    constexpr int PMAX = (NCHUNK + W - 1) / W;
    constexpr int NB = 2;
    constexpr int BS = (PMAX + NB - 1) / NB;
    constexpr int HALF = STEPS / 2;
    constexpr int LAG = (HALF > 3) ? HALF - 2 : 1;
    constexpr int PSTRIDE = W * 1024;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int half = lane >> 5;
    const int64_t wrow0 = (int64_t)blockIdx.x * VQ_SCREEN_F32_ROWS + wave * 32;

    const int my_pieces = (NCHUNK - wave + W - 1) / W;
    const int piece_off = wave * 1024 + lane * 16;
    for (int k = 0; k < my_pieces; ++k)
        *(f32x4 *)(smem + piece_off + k * PSTRIDE) = *(const f32x4 *)(a.tiles + piece_off + (size_t)k * PSTRIDE);

    // ---- x rows -> two bf16 B-operand sets; lane (j, half) holds features 16 ks + 8 half + 0..7 ----
    const int64_t row = wrow0 + j;
    const bool row_ok = row < a.N;
    uint4 xh[NK], xm[NK];
    float eps;
    {
        const float *p = (const float *)a.x + (row_ok ? row : (a.N - 1)) * a.ldx + 8 * half;
        float xs = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const f32x4 v0 = *(const f32x4 *)(p + ks * 16), v1 = *(const f32x4 *)(p + ks * 16 + 4);
            split8_bf16(v0, v1, xh[ks], xm[ks]);
            xs = __builtin_fmaf(v0.x, v0.x, xs); xs = __builtin_fmaf(v0.y, v0.y, xs);
            xs = __builtin_fmaf(v0.z, v0.z, xs); xs = __builtin_fmaf(v0.w, v0.w, xs);
            xs = __builtin_fmaf(v1.x, v1.x, xs); xs = __builtin_fmaf(v1.y, v1.y, xs);
            xs = __builtin_fmaf(v1.z, v1.z, xs); xs = __builtin_fmaf(v1.w, v1.w, xs);
        }
        xs += __shfl_xor(xs, 32, 64);
        xs *= 1.001f;
        const float y2max = __uint_as_float(a.scalars[0]);
        const float xy = sqrtf(xs) * 1.0001f * sqrtf(y2max) * 1.0001f;
        const float u = 5.9604645e-8f;   // 2^-24
        if (METRIC == 0) eps = u * (10.f * (xs + y2max + 2.f * xy) + (2.f * DT + 1600.f) * xy + (12.f * DT + 2.f) * (xy + 0.5f * y2max)) + 4e-8f;
        else             eps = 2.f * u * (7.f * DT + 802.f) * xy + 1e-30f;
    }

    float m1 = -__builtin_inff(), m2 = -__builtin_inff();
    int tix = 0;
    const int nt = a.n_tiles;
    for (int ct = 0; ct < nt; ++ct) {
        const int buf = ct & 1;
        __syncthreads();
        const char *tile = smem + buf * TILE_B;
        const bool more = ct + 1 < nt;
        const char *gsrc = a.tiles + (size_t)(more ? ct + 1 : ct) * TILE_B + piece_off;
        char *ldst = smem + (buf ^ 1) * TILE_B + piece_off;
        const int npieces = more ? my_pieces : 0;

        const float *nh = (const float *)(tile + 128 * DT);
        f32x16 acc0, acc1;
        if (METRIC == 0 || (ct == nt - 1 && (a.C & 31))) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *(const f32x4 *)(nh + 8 * q + 4 * half);
                if (METRIC != 0) { v.x = v.x < -1e38f ? v.x : 0.f; v.y = v.y < -1e38f ? v.y : 0.f;
                                   v.z = v.z < -1e38f ? v.z : 0.f; v.w = v.w < -1e38f ? v.w : 0.f; }
                acc0[4 * q + 0] = v.x; acc0[4 * q + 1] = v.y; acc0[4 * q + 2] = v.z; acc0[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;

        const uint4 *ap = (const uint4 *)tile + lane;
        f32x4 stg[BS];
        uint4 af[VQS_PF];
#pragma unroll
        for (int p = 0; p < VQS_PF; ++p) af[p] = ap[(p < STEPS ? p : 0) * 64];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int ks = s >> 1;
            const bf16x8 av = __builtin_bit_cast(bf16x8, af[s % VQS_PF]);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, xh[ks]);
            // accumulators strictly alternate over the 3 MFMAs of a k-step (k-step parity picks who starts)
            if ((s & 1) == 0) {        // c_hi: times x_hi and x_mid
                const bf16x8 bm = __builtin_bit_cast(bf16x8, xm[ks]);
                if ((ks & 1) == 0) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bm, acc1, 0, 0, 0);
                } else {
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bm, acc0, 0, 0, 0);
                }
            } else {                   // c_lo: times x_hi
                if ((ks & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, acc0, 0, 0, 0);
                else               acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, acc1, 0, 0, 0);
            }
            if (s + VQS_PF < STEPS) af[s % VQS_PF] = ap[(s + VQS_PF) * 64];
#ifdef VQS_PIN
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (s == b * HALF) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < BS; ++i)
                        if (b * BS + i < PMAX) stg[i] = *(const f32x4 *)(gsrc + (size_t)(b * BS + i) * PSTRIDE);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s == b * HALF + LAG) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < BS; ++i)
                        if (b * BS + i < npieces) *(f32x4 *)(ldst + (b * BS + i) * PSTRIDE) = stg[i];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] += acc1[r];
        top2_tile(acc0, m1, m2, tix, ct);
    }

    // ---- merge the half-waves, certify, emit ----
    int code;
    bool flagged;
    {
        const int e = (int)(__float_as_uint(m1) & 15u);
        const int c_own = tix * 32 + 8 * (e >> 2) + 4 * half + (e & 3);
        const float o1 = __shfl_xor(m1, 32, 64);
        const float o2 = __shfl_xor(m2, 32, 64);
        const int oc = __shfl_xor(c_own, 32, 64);
        const bool take = o1 > m1;
        const float b1 = take ? o1 : m1;
        const float lo1 = take ? m1 : o1;
        const float b2 = fmaxf(lo1, fmaxf(m2, o2));
        code = take ? oc : c_own;
        const float thr = eps + 8e-6f * fabsf(b1);
        flagged = !((b1 - b2) > thr) || code >= a.C;
        if (code >= a.C) code = 0;
        if (row_ok && half == 0) {
            a.idx_out[row] = (int64_t)code;
            if (a.dbg) {
                float *d = a.dbg + row * 4;
                d[0] = b1; d[1] = b2; d[2] = thr; d[3] = flagged ? 1.f : 0.f;
            }
        }
        const bool f = flagged && row_ok && half == 0;
        const unsigned long long bal = __ballot(f);
        if (bal) {
            int base = 0;
            if (lane == 0) base = atomicAdd(a.flag_count, (int)__popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (f) {
                const int slot = base + (int)__popcll(bal & ((1ull << lane) - 1ull));
                a.flag_rows[slot] = (int)row;
                a.flag_keys[slot] = ~0ull;
            }
        }
    }

    // ---- q rows (fp32, from embed) and squared error: whole rows per wave, 8 in flight; x is re-read (coalesced) ----
    if (a.q_out || a.sqerr_partial || a.resid_out) {
        const int counted = (row_ok && !flagged && (!a.row_mask || a.row_mask[row] != 0)) ? 1 : 0;
        const bool want_sq = a.sqerr_partial != nullptr;
        const bool want_x = want_sq || a.resid_out != nullptr;
        double ds = 0.0;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 8) {
            f32x4 g[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = __builtin_amdgcn_readlane(code, r0 + u);
                const int64_t rr = wrow0 + r0 + u;
                if (lane * 4 < DT) {
                    g[u] = *(const f32x4 *)(a.embed + (size_t)c * DT + lane * 4);
                    if (want_x) xv[u] = *(const f32x4 *)((const float *)a.x + (rr < a.N ? rr : a.N - 1) * a.ldx + lane * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t rr = wrow0 + r0 + u;
                const int cnt = __builtin_amdgcn_readlane(counted, r0 + u);
                if (lane * 4 < DT) {
                    if (a.q_out && rr < a.N) *(f32x4 *)((float *)a.q_out + rr * a.ldq + lane * 4) = g[u];
                    if (a.resid_out && rr < a.N) *(f32x4 *)((float *)a.resid_out + rr * a.ldr + lane * 4) = xv[u] - g[u];
                    if (want_sq && cnt) {
                        const float d0 = g[u].x - xv[u].x, d1 = g[u].y - xv[u].y, d2 = g[u].z - xv[u].z, d3 = g[u].w - xv[u].w;
                        ds += (double)(((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3);
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
    // 16-byte header (count) | N ints (row list, padded to 8 bytes) | N u64 (keys of the exact pass)
    return N <= 0 ? 0 : 16 + (((size_t)N * sizeof(int) + 7) & ~(size_t)7) + (size_t)N * sizeof(unsigned long long);
}

extern "C" int vqhip_screen_supported(int64_t N, int D, int C)
{
    return (D == 32 || D == 64 || D == 128 || D == 256) && N > 0 && N < ((int64_t)1 << 31) - 512 && C >= 2;
}

template <int DT, int METRIC>
static int launch_screen(const ScreenArgs &a, int x_dtype, hipStream_t st)
{
    constexpr int SMEM = 2 * (128 * DT + 1024);
    static VqAttrOnce once_b, once_f;
    if (int rc = vq_set_max_smem(once_b, (const void *)vq_screen_kernel<DT, METRIC>, SMEM, "vq_screen_kernel")) return rc;
    if (int rc = vq_set_max_smem(once_f, (const void *)vq_screen_f32_kernel<DT, METRIC>, SMEM, "vq_screen_f32_kernel")) return rc;
    const unsigned blocks = (unsigned)vqhip_screen_blocks(a.N, x_dtype);
    if (x_dtype == VQHIP_BF16)
        hipLaunchKernelGGL((vq_screen_kernel<DT, METRIC>), dim3(blocks), dim3(VQS_WAVES * 64), SMEM, st, a);
    else
        hipLaunchKernelGGL((vq_screen_f32_kernel<DT, METRIC>), dim3(blocks), dim3(VQS_F32_WAVES * 64), SMEM, st, a);
    return vq_launch_status("vq_screen_kernel");
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
    if (N < 0 || C <= 0) VQ_FAIL(VQHIP_EINVAL, "assign_screened: N < 0 or C <= 0");
    if (N == 0) return 0;
    if (!x || !packed || !embed || !idx_out || !workspace) VQ_FAIL(VQHIP_EINVAL, "assign_screened: null pointer");
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "assign_screened: unknown x dtype %d", x_dtype);
    if (metric != VQHIP_EUCLID && metric != VQHIP_COSINE_PRENORM)
        VQ_FAIL(VQHIP_EINVAL, "assign_screened: metric %d (VQHIP_EUCLID, or VQHIP_COSINE_PRENORM on rows normalised by vqhip_l2norm_rows)", metric);
    if (!vqhip_screen_supported(N, D, C)) VQ_FAIL(VQHIP_EDIM, "assign_screened: N=%lld D=%d C=%d outside the screened path (D in {32,64,128,256}, C >= 2)", (long long)N, D, C);
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
    int *rows = count + 4;
    hipError_t e = hipMemsetAsync(count, 0, 16, st);
    if (e != hipSuccess) VQ_FAIL((int)e, "assign_screened: hipMemsetAsync: %s", hipGetErrorString(e));

    const char *base = (const char *)packed;
    ScreenArgs a;
    a.x = x; a.N = N; a.ldx = ldx;
    a.tiles = base + vq_packed_screen_offset(C, D);
    a.embed_bf16 = (const unsigned short *)(base + vq_packed_bf16_offset(C, D));
    a.embed = embed;
    a.scalars = (const unsigned *)(base + vq_packed_scalars_offset(C, D));
    a.C = C; a.n_tiles = (C + 31) / 32;
    a.idx_out = idx_out; a.q_out = q_out; a.ldq = ldq; a.resid_out = resid_out; a.ldr = ldr;
    a.sqerr_partial = sqerr_partial; a.row_mask = row_mask;
    unsigned long long *keys = (unsigned long long *)((char *)workspace + 16 + (((size_t)N * sizeof(int) + 7) & ~(size_t)7));
    a.flag_count = count; a.flag_rows = rows; a.flag_keys = keys; a.dbg = debug_out;
#ifdef VQ_TRACE
    a.trace = vq_g_trace;
#endif
    int rc;
    switch (D) {
        case 32: rc = dispatch_screen<32>(a, x_dtype, metric, st); break;
        case 64: rc = dispatch_screen<64>(a, x_dtype, metric, st); break;
        case 128: rc = dispatch_screen<128>(a, x_dtype, metric, st); break;
        default: rc = dispatch_screen<256>(a, x_dtype, metric, st); break;
    }
    if (rc) return rc;
    return vq_assign_listed(x, x_dtype, metric, N, D, ldx, packed, embed, C, idx_out, q_out, ldq, resid_out, ldr,
                            sqerr_partial ? sqerr_partial + vqhip_screen_blocks(N, x_dtype) : nullptr, row_mask, rows, count, keys, st);
}
