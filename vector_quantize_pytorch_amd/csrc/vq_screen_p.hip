// vq_screen_p.hip -- persistent, software-pipelined form of the single-pass fp16 screen (vq_screen16_kernel in vq_screen.hip),
// gfx950 only.  Same certificate, same outputs (index, q rows, open / pair lists); what changes is WHEN things happen:
//
//   * ONE workgroup of 4 waves per CU -- ONE wave per SIMD, which therefore owns the SIMD's whole 512-entry register file -- and
//     every wave loops over 64-row pieces (two 32-row blocks) of "super-blocks" of 256 consecutive rows (workgroup b takes
//     sb = b, b + G, b + 2 G, ...).  A lone wave keeps its matrix pipe busy only if what it issues BESIDE an MFMA fits the MFMA's
//     shadow (32 cycles = 8 issue slots): the sweep is the paired sweep of vq_screen16_kernel (one A fragment from LDS -> two MFMAs
//     on different accumulators, the previous tile's top-3 fold between them), and everything else rides in the same stream:
//       - the next super-block's raw bf16 rows are requested during the first two intervals of the sweep into registers of their
//         own (128: the register file has room for the current AND the next rows), their norms are summed and the scale agreed
//         mid-sweep, and they are converted into the operand registers during the LAST tile, k-step by k-step, behind the last
//         MFMA that reads each operand;
//       - the previous super-block's q rows (gather of bf16 code rows from L2 -> store) move four rows per interval;
//       - its open / pair list entries are written once their atomics have returned.
//     No phase of a workgroup is memory-only: HBM traffic is spread over the whole kernel, and there is no second workgroup whose
//     memory phase slows the sweep (vq_screen16_kernel: 4.8 k cycles per interval next to a neighbour's output phase, 3.4 k alone).
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
#define VQP_MIN_TILES 32          // the schedule of a block (q rows in intervals 0 .. 15) needs 16 intervals

template <int DT> struct ScreenPCfg {
    static constexpr int W = 4;                        // waves per workgroup: one per SIMD
    static constexpr int NK = DT / 16;                 // MFMA k-steps per tile
    static constexpr int TILE_B = 64 * DT + 1024;      // fp16 tile: 32 codes x DT + 32 floats -||c||^2/2 (padded to 1 KiB)
    static constexpr int SUB = 2;                      // tiles per barrier interval
    static constexpr int SUPER_B = SUB * TILE_B;
    static constexpr int NCHUNK = SUPER_B / 1024;
    static constexpr int PMAX = (NCHUNK + W - 1) / W;  // 1-KiB pieces per wave and interval
    static constexpr int BUF_B = PMAX * W * 1024;      // LDS bytes per buffer (every wave copies PMAX pieces, unconditionally)
    static constexpr int NB = 3;                       // LDS buffers: interval I + 1 is staged during interval I, and the barrier of an
                                                       // interval comes BEFORE its last k-steps (so that the next interval's first
                                                       // fragments are requested behind it): the buffer being written was last read
                                                       // two barriers ago
    static size_t smem_bytes(int n_tiles16) { return NB * (size_t)BUF_B + 2 * (size_t)n_tiles16 * 32 * 4 + 256; }
};

__device__ __forceinline__ void vqp_barrier()
{
    // LDS-only barrier: the ds_writes of this wave have landed (lgkmcnt), global loads stay in flight across it
#ifdef VQP_BARE_BARRIER
    asm volatile("s_barrier" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

template <int DT, int METRIC, bool HASQ>
__global__ void __launch_bounds__(256, 1) vq_screenp_kernel(const ScreenArgs a, const int nsb, long long *const trace)
{
#ifdef VQP_TRACE            // dev build: s_memtime stamps of the first VQP_TRACE_IV intervals of workgroups 0 .. 7 (tools/trace_screenp.py)
#define VQP_TRACE_IV 48
    int trace_gi = 0;
#define VQP_STAMP(k) do { if (trace && blockIdx.x < 8 && trace_gi < VQP_TRACE_IV && lane == 0) \
        trace[(((size_t)blockIdx.x * 8 + wave) * VQP_TRACE_IV + trace_gi) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define VQP_STAMP_END(k) do { if (trace && blockIdx.x < 8 && trace_gi <= VQP_TRACE_IV && lane == 0) \
        trace[(((size_t)blockIdx.x * 8 + wave) * VQP_TRACE_IV + trace_gi - 1) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define VQP_STAMP(k) do {} while (0)
#define VQP_STAMP_END(k) do {} while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Cfg = ScreenPCfg<DT>;
    constexpr int NK = Cfg::NK, TILE_B = Cfg::TILE_B, SUB = Cfg::SUB, SUPER_B = Cfg::SUPER_B;
    constexpr int PMAX = Cfg::PMAX, BUF_B = Cfg::BUF_B, NB = Cfg::NB;
#ifndef VQP_PF
#define VQP_PF 4
#endif
    constexpr int PF = VQP_PF;                         // A-fragment ring depth
    static_assert(NK % PF == 0, "the ring runs on across tile boundaries: k-step s of every tile sits in slot s % PF");
    static_assert(SUB == 2, "an interval is two tiles: the accumulator sets swap roles inside it");

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
    float *const nrm = (float *)(smem + NB * BUF_B);   // [2][CP] start values, scaled for the current / next iteration
    unsigned *const xch = (unsigned *)(nrm + 2 * CP); // [2][8] largest finite ||x||^2 bits of every wave
    const char *const tiles = a.tiles16;
    (void)nt16_bytes;
    const int sc = (int)a.scalars[2];
    const float y2max = __uint_as_float(a.scalars[0]);
    const float rmax = __uint_as_float(a.scalars[1]);
    const float ymax = sqrtf(y2max) * 1.0001f;
    const unsigned lane16_0 = (unsigned)lane * 16u;
    const unsigned wc_0 = (unsigned)wave * (unsigned)(PMAX * 1024) + lane16_0;   // ... inside the wave's contiguous PMAX-KiB share of a buffer

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
    auto wg_max = [&](const unsigned *p) -> unsigned { // max of the 4 waves' entries (uniform LDS read)
        const uint4 u0 = *(const uint4 *)p;
        return (unsigned)__builtin_amdgcn_readfirstlane((int)max(max(u0.x, u0.y), max(u0.z, u0.w)));
    };
    auto wave_max_u = [&](unsigned v) -> unsigned {     // DPP inside the rows of 16 + four readlanes (an LDS atomicMax from all 64 lanes
                                                        // becomes a 64-trip scalar loop: hipcc's atomic optimizer)
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
        v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
        const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
        const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
        return max(max(r0, r1), max(r2, r3));
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
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));                 // (the four 64-bit addresses per trip are loop invariants of the block loop otherwise)
#pragma clang loop vectorize(disable) unroll(disable)
        for (int i0 = tid_; i0 < CP; i0 += 1024) {     // CP is a multiple of 128: four values per thread and trip, requested together
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + 256 * u, CP - 1);
                v[u] = *(const float *)(tiles + (size_t)(i >> 5) * TILE_B + 64 * DT + (i & 31) * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 256 * u;
                if (METRIC != 0) v[u] = v[u] < -1e38f ? v[u] : 0.f;
                if (i < CP) dst[i] = fmaxf(v[u] * SS, -3.0e38f);
            }
        }
    };
    auto row_norm2 = [&](const uint4 (&r)[NK]) -> float {   // ||x||^2 of this lane's half of a row (any order: it only scales the bound)
        float xs[4] = {0.f, 0.f, 0.f, 0.f};            // four chains: one v_dot2c accumulating into itself 64 times waits for itself
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const unsigned w[4] = {r[ks].x, r[ks].y, r[ks].z, r[ks].w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                xs[q] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[q]), __builtin_bit_cast(bf16x2, w[q]), xs[q], false);
        }
        return (xs[0] + xs[1]) + (xs[2] + xs[3]);
    };

    // ---- registers that live across the whole kernel ---------------------------------------------------------------------------
    // lane (j, half) of a wave: row j of both of the wave's 32-row blocks, elements 16 ks + 8 half + 0..7 of every k-step
    uint4 cur[2][NK], nxt[2][NK];                      // B operands of the piece being swept / raw rows of the next one
    f32x16 accA[2], accB[2];                           // two accumulator sets x two row blocks
    float m1[2], m2[2], m3[2];                         // top 3 per row block (independent chains, interleaved instruction by instruction)
    int tix[2], tix2[2];

    // state of the piece being swept
    float eps_c[2], SS_c, iSS_c;
    // state of the next piece (built by the fillers)
    float xs2_n[2] = {0.f, 0.f};
    int SX_n = 0;
    // outputs of the previous piece still to be written (q rows, list entries); PACKED: lane l <-> row l of the wave's 64 rows
    bool has_prev = false;
    int pcode = 0, pcls = 0, pid2 = 0, prow = 0, pbase_o = 0, pbase_p = 0;
    unsigned long long pbalo = 0ull, pbalp = 0ull;
    int plim = 0;                                      // last valid row of the previous piece, relative to its first (clamped to 63)
    char *pqbase = (char *)a.q_out;                    // q_out + first row * ldq * 2 (wave-uniform)
    const int ldq2 = (int)(a.ldq * 2);
    const int ldx2 = (int)(a.ldx * 2);
    const unsigned j16 = (unsigned)j * 16u;

    // the rows of super-block `s`, row block rb: wave-uniform base (SGPRs) + 32-bit lane offset; rows past the end repeat the last one
    // (part: -1 all 16 pieces of the row, 0 / 1 its first / second half -- the CU's address unit needs ~2 k cycles for the 64 KiB of a
    //  super-block in this one-row-per-lane pattern: eight loads per wave and interval keep that in the background)
    auto load_rows = [&](int s, int rb, uint4 (&dst)[NK], int part = -1) __attribute__((always_inline)) {
        const int64_t r0 = (int64_t)s * 256 + wave * 64 + rb * 32;
        const int64_t lim = a.N - 1 - r0;
        const char *const xb = (const char *)a.x + (r0 < a.N ? r0 : a.N - 1) * a.ldx * 2;
        unsigned xo = (unsigned)min(j, lim > 31 ? 31 : (lim < 0 ? 0 : (int)lim)) * (unsigned)ldx2 + (unsigned)half * 16u;
        asm volatile("" : "+v"(xo));                   // formed here (hoisted out of the interval loop, the addresses cost VGPR pairs)
        const char *const xl = xb + xo;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
            if (part < 0 || ks / (NK / 2) == part) dst[ks] = *(const uint4 *)(xl + ks * 32);
    };

    // ---- prologue: first piece's rows, first codebook buffer, scale, start values ------------------------------------------------
    int sb = blockIdx.x;
#ifndef VQP_STAGGER
#define VQP_STAGGER 8
#endif
    // Every workgroup does the same work at the same pace: started together, all 256 of them would request their next rows, append to
    // the lists and write their outputs in the same microsecond of every block period.  The start is spread over about one block
    // period instead (the workgroups of one XCD -- blockIdx mod 8 -- get phases from all over the period).
    for (int i = ((int)(blockIdx.x >> 3) & 31) * VQP_STAGGER; i > 0; --i) __builtin_amdgcn_s_sleep(2);
    {
        load_rows(sb, 0, nxt[0]);
        load_rows(sb, 1, nxt[1]);
#pragma unroll
        for (int k = 0; k < PMAX; ++k)                   // interval 0 -> buffer 0
            *(f32x4 *)(smem + wc_0 + k * 1024) = *(const f32x4 *)(tiles + wc_0 + (size_t)k * 1024);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float xs = row_norm2(nxt[rb]);
            xs += xor32f(xs);
            xs2_n[rb] = xs * 1.001f;
        }
        {
            const unsigned wm = wave_max_u(max(finite_bits(xs2_n[0]), finite_bits(xs2_n[1])));
            if (lane == 0) xch[wave] = wm;
        }
        vqp_barrier();
        SX_n = pick_sx(wg_max(xch));
        const float S = __uint_as_float((unsigned)(SX_n + 127) << 23);
        scale_norms(nrm, __uint_as_float((unsigned)(SX_n + sc + 127) << 23));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < NK; ++ks)
                cur[rb][ks] = make_uint4(conv_word(nxt[rb][ks].x, S), conv_word(nxt[rb][ks].y, S), conv_word(nxt[rb][ks].z, S), conv_word(nxt[rb][ks].w, S));
        SS_c = __uint_as_float((unsigned)(SX_n + sc + 127) << 23);
        iSS_c = __uint_as_float((unsigned)(127 - SX_n - sc) << 23);
        eps_c[0] = eps_of(xs2_n[0], SX_n);
        eps_c[1] = eps_of(xs2_n[1], SX_n);
        vqp_barrier();                                  // buffer 0 and the start values are in LDS for every wave
    }

    // one score into (best, second, third) of its row block: the key carries the register number in its 4 low mantissa bits.  ONE asm
    // statement per score (as builtins hipcc sinks every v_med3 of a tile behind the tile's last MFMA, vq_screen.hip); it sits in the
    // shadow of ONE MFMA: with a single wave on the SIMD nothing else fills the 32 cycles between two matrix instructions
    auto fold1 = [&](const f32x16 &P, int e, int t) __attribute__((always_inline)) {
        float k;
        asm volatile("v_and_or_b32 %3, %4, -16, %5\n\tv_med3_f32 %2, %1, %2, %3\n\tv_med3_f32 %1, %0, %1, %3\n\tv_max_f32 %0, %0, %3"
                     : "+v"(m1[t]), "+v"(m2[t]), "+v"(m3[t]), "=&v"(k) : "v"(P[e]), "n"(e));
    };
    auto book = [&](int t, float om1, float om2, int tile_id) __attribute__((always_inline)) {   // which tiles hold best / second of block t
        const bool c1 = m1[t] != om1;
        const int from_old_best = (c1 && m2[t] == om1) ? tix[t] : tile_id;
        tix2[t] = (m2[t] != om2) ? from_old_best : tix2[t];
        tix[t] = c1 ? tile_id : tix[t];
        // pinned here: hipcc otherwise sinks every tile's bookkeeping to the end of the block and keeps a copy of (best, second)
        // per tile alive until then
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

    // carried across tiles, intervals and blocks: the A-fragment ring and the start values of the NEXT tile, the LDS buffer in use
    uint4 af[PF];
    int bcur = 0;
    {
        const uint4 *ap = (const uint4 *)(smem + lane16_0);
        const float *nh = nrm + 4 * half;
#pragma unroll
        for (int p = 0; p < PF; ++p) af[p] = ap[p * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *(const f32x4 *)(nh + 8 * q);
            accA[0][4 * q + 0] = v.x; accA[0][4 * q + 1] = v.y; accA[0][4 * q + 2] = v.z; accA[0][4 * q + 3] = v.w;
        }
    }

    // the special intervals of a block (the schedule is the same for every codebook size, nst >= 16)
    const int I_norm = nst >> 1;                       // ||x||^2 of the next piece's rows (requested at intervals 0, 1) + scale exchange
    // ... at I_norm + 2: scale agreed (>= one barrier later) + start values;  at nst - 1: the rows are converted

    auto reset_fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) { m1[t] = m2[t] = m3[t] = -__builtin_inff(); tix[t] = tix2[t] = 0; }
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[0][r] = accB[1][r] = -3.0e38f;   // "padding codes" in front of the first tile: never win
    };
    reset_fold();
    for (int it = 0;; ++it, sb += G) {
        const bool has_next = sb + G < nsb;
        const int sb_n = has_next ? sb + G : nsb - 1;
        const float *const nrm_c = nrm + (it & 1) * CP + 4 * half;        // this iteration's start values (+ this half's 4 codes)
        const float *const nrm_nx = nrm + ((it + 1) & 1) * CP + 4 * half; // the next iteration's (its first tile is prefetched from here)
        float *const nrm_n = nrm + ((it + 1) & 1) * CP;
        unsigned *const xch_n = xch + ((it + 1) & 1) * 8;

        int ptile = 0;

        // one barrier interval = two tiles.  KIND selects the block-level work that rides in it: 0 nothing, 1 / 6 request the next
        // piece's rows (row block 0 / 1), 2 list entries of the previous piece, 3 row norms + scale exchange, 4 agree the scale +
        // start values, 7 convert the rows into the operand registers (last interval).
        // (Separate instantiations instead of run-time branches inside one loop: a branch that redefines the `nxt` registers
        //  makes hipcc shuffle all of them through copies at the join of EVERY interval.)
        auto interval = [&](const int I, auto Kc) __attribute__((always_inline)) {
            constexpr int KIND = decltype(Kc)::value;
            VQP_STAMP(0);
            // 16 x lane, re-derived once per interval instead of living in registers across the kernel: every lane-dependent LDS / tile
            // address below is this value plus something wave-uniform, and hipcc otherwise hoists each of those sums out of the block
            // loop into a register of its own (spilled to scratch: a reload costs a vmcnt(0))
            unsigned lane16;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(lane16));
            const unsigned wc = (unsigned)wave * (unsigned)(PMAX * 1024) + lane16;
            const char *const sbase = smem + bcur * BUF_B;
            const int bnext = bcur + 1 == NB ? 0 : bcur + 1;
            const char *const sbase_n = smem + bnext * BUF_B;
            int siv = I + 1;
            siv = siv >= nst ? 0 : siv;
            const unsigned gsrc = (unsigned)siv * (unsigned)SUPER_B;   // byte offset of the staged interval's tiles (SGPR: soffset)
            char *const ldst_w = smem + bnext * BUF_B + wave * (PMAX * 1024);   // wave-uniform: the DMA adds 16 x lane itself
            [[maybe_unused]] uint4 gq0, gq1;
            const bool q_iv = HASQ && I < 16;
            const float Sn = __uint_as_float((unsigned)(SX_n + 127) << 23);

            // one tile: accumulate into C, fold the previous tile's P.  The A-fragment ring `af` and the start values `init` arrive
            // loaded for this tile's first k-steps and leave loaded for the next tile's (apn / nhn: where those live)
            auto tile_body = [&](auto Sub, f32x16 (&C)[2], f32x16 (&P)[2]) __attribute__((always_inline)) {
                constexpr int sub = decltype(Sub)::value;
                const int tile_id = 2 * I + sub;
                const uint4 *ap = (const uint4 *)(sbase + sub * TILE_B + lane16);
                const uint4 *apn = sub == 0 ? (const uint4 *)(sbase + TILE_B + lane16) : (const uint4 *)(sbase_n + lane16);
                int tn = tile_id + 1;
                tn = tn >= nt16 ? 0 : tn;
                const float *nhn = (sub == 0 || tn != 0 ? nrm_c : nrm_nx) + tn * 32;
                const float o10 = m1[0], o20 = m2[0], o11 = m1[1], o21 = m2[1];
#pragma unroll
                for (int s_ = 0; s_ < NK; ++s_) {
                    const f16x8 av = __builtin_bit_cast(f16x8, af[s_ % PF]);
                    // C[0] arrives holding the start values -||c||^2/2 (scaled; loaded into it behind the fold of its last use): the first
                    // MFMA of the tile takes them as its C operand into C[1], the second accumulates in place
#ifndef VQP_NO_MFMA
                    C[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, cur[1][s_]), s_ == 0 ? C[0] : C[1], 0, 0, 0);
#else
                    if (s_ == 0) C[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, cur[1][s_]), C[0], 0, 0, 0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- gap A ----
#ifndef VQP_NO_FOLD
                    fold1(P[1], s_, 1);
#else
                    if (s_ == 0) fold1(P[1], s_, 1);
#endif
                    __builtin_amdgcn_sched_barrier(0);
#ifndef VQP_NO_MFMA
                    C[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, cur[0][s_]), C[0], 0, 0, 0);
#else
                    if (s_ == 0) C[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, cur[0][s_]), C[0], 0, 0, 0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- gap B ----
                    if (sub == 1 && s_ == NK - PF - 1) {
                        VQP_STAMP(1);
                        // this wave's DMA pieces (requested a tile ago) have landed: everything but the memory instructions issued AFTER
                        // them in this interval -- the 8 row loads of KIND 1 / 6 / 8 / 9 (HBM latency: not waited for here), the q rows' two
                        // gathers and two stores -- has retired (vmcnt retires in order)
#ifdef VQP_VMCNT0
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
                        if (q_iv && has_prev) {
                            if constexpr (KIND == 1 || KIND == 6 || KIND == 8 || KIND == 9) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                            else                                  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        } else {
                            if constexpr (KIND == 1 || KIND == 6 || KIND == 8 || KIND == 9) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                            else                                  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
#endif
                        vqp_barrier();                   // every wave's share of the next interval's tiles is in LDS
                        VQP_STAMP(2);
                    }
                    if (s_ + PF < NK) af[s_ % PF] = ap[(s_ + PF) * 64];
                    else              af[s_ % PF] = apn[(s_ + PF - NK) * 64];
#ifndef VQP_NO_FOLD
                    fold1(P[0], s_, 0);
#else
                    if (s_ == 0) fold1(P[0], s_, 0);
#endif
                    if (s_ >= NK - 4) {                  // the next tile's start values into P[0]: registers 4 q .. 4 q + 3 were folded by k-step 4 q + 3
                        const int q = s_ - (NK - 4);
                        const f32x4 v = *(const f32x4 *)(nhn + 8 * q);
                        P[0][4 * q + 0] = v.x; P[0][4 * q + 1] = v.y; P[0][4 * q + 2] = v.z; P[0][4 * q + 3] = v.w;
                    }
                    if constexpr (KIND == 7 && sub == 1) {
                        // the last tile of the block: operand s_ has been read for the last time -> the next piece's rows take its place
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) {
                            cur[rb][s_] = make_uint4(conv_word(nxt[rb][s_].x, Sn), conv_word(nxt[rb][s_].y, Sn),
                                                     conv_word(nxt[rb][s_].z, Sn), conv_word(nxt[rb][s_].w, Sn));
                            // pinned HERE (its consumer is the next block's first MFMA: hipcc sinks the whole conversion down there)
#ifndef VQP_NO_PIN
                            asm volatile("" : "+v"(cur[rb][s_].x), "+v"(cur[rb][s_].y), "+v"(cur[rb][s_].z), "+v"(cur[rb][s_].w));
#endif
                        }
                    }
                    if (sub == 0 && s_ == 0) {
#ifndef VQP_NO_STAGE
                        // LDS-DMA: 1 KiB per instruction straight into the next buffer (lane-linear image = the packed tiles' own
                        // layout), no staging registers, no ds_write pass; retired by the vmcnt(0) in front of this interval's barrier
                        // (asm, not __builtin_amdgcn_global_load_lds: hipcc follows every DMA it knows of with a vmcnt(0) before the next
                        //  ds_read -- the DMA writes LDS -- which serialises the L2 round trip into every interval)
                        const char *const gp = tiles + gsrc;                          // wave-uniform: SGPR pair
                        const unsigned lb = (unsigned)(uintptr_t)ldst_w;
#ifdef VQP_GLDS_OFFSET   // the instruction offset moves source and destination alike
                        static_assert(PMAX == 9, "three groups of pieces");
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                                     "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072\n\t"
                                     :: "v"(wc), "s"(gp), "s"(lb) : "memory");
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                                     "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072\n\t"
                                     :: "v"(wc), "s"(gp + 4096), "s"(lb + 4096u) : "memory");
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                                     :: "v"(wc), "s"(gp + 8192), "s"(lb + 8192u) : "memory");
#else
#pragma unroll
                        for (int i = 0; i < PMAX; ++i)
                            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                                         :: "v"(wc), "s"(gp + i * 1024), "s"(lb + (unsigned)i * 1024u) : "memory");
#endif
#endif
                    }
                    if constexpr (KIND == 1 && sub == 0) { if (s_ == 1) load_rows(sb_n, 0, nxt[0], 0); }
                    if constexpr (KIND == 6 && sub == 0) { if (s_ == 1) load_rows(sb_n, 0, nxt[0], 1); }
                    if constexpr (KIND == 8 && sub == 0) { if (s_ == 1) load_rows(sb_n, 1, nxt[1], 0); }
                    if constexpr (KIND == 9 && sub == 0) { if (s_ == 1) load_rows(sb_n, 1, nxt[1], 1); }
                    if (sub == 0 && s_ == 2 && q_iv) {  // q rows of the previous piece: rows 4 I .. 4 I + 3, one half-wave each
                        const int c0 = __builtin_amdgcn_ds_bpermute((4 * I + half) * 4, pcode);
                        const int c1 = __builtin_amdgcn_ds_bpermute((4 * I + 2 + half) * 4, pcode);
                        gq0 = *(const uint4 *)((const char *)a.embed_bf16 + ((unsigned)c0 * (unsigned)(DT * 2) + j16));
                        gq1 = *(const uint4 *)((const char *)a.embed_bf16 + ((unsigned)c1 * (unsigned)(DT * 2) + j16));
                    }
                    if (sub == 1 && s_ == 6 && q_iv && has_prev) {
                        // rows past the end repeat the last row (same code, same bytes): row offset clamped to plim
                        const int r0 = min(4 * I + half, plim), r1 = min(4 * I + 2 + half, plim);
                        *(uint4 *)(pqbase + ((unsigned)(r0 * ldq2) + j16)) = gq0;
                        *(uint4 *)(pqbase + ((unsigned)(r1 * ldq2) + j16)) = gq1;
                    }
                    __builtin_amdgcn_sched_barrier(0);  // pins the slices and the prefetch distance between the MFMAs
                }
                book(0, o10, o20, ptile);
                book(1, o11, o21, ptile);
                ptile = tile_id;
            };
            tile_body(std::integral_constant<int, 0>{}, accA, accB);
            tile_body(std::integral_constant<int, 1>{}, accB, accA);
            bcur = bnext;

            // ---- the special intervals ----
            if constexpr (KIND == 2) {
                if (has_prev) write_lists(pbalo, pbalp, pbase_o, pbase_p, pcls, prow, pcode, pid2);   // (its atomics have returned)
            } else if constexpr (KIND == 3) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    float xs = row_norm2(nxt[rb]);
                    xs += xor32f(xs);
                    xs2_n[rb] = xs * 1.001f;
                }
                const unsigned wm = wave_max_u(max(finite_bits(xs2_n[0]), finite_bits(xs2_n[1])));
                if (lane == 0) xch_n[wave] = wm;         // read two barriers later (KIND 4)
            } else if constexpr (KIND == 4) {
                SX_n = pick_sx(wg_max(xch_n));
                scale_norms(nrm_n, __uint_as_float((unsigned)(SX_n + sc + 127) << 23));
            }
            VQP_STAMP(3);
#ifdef VQP_TRACE
            ++trace_gi;
#endif
        };
        using K0 = std::integral_constant<int, 0>;
        interval(0, std::integral_constant<int, 1>{});
        interval(1, std::integral_constant<int, 6>{});
        interval(2, std::integral_constant<int, 2>{});
        interval(3, std::integral_constant<int, 8>{});
        interval(4, std::integral_constant<int, 9>{});
#pragma unroll 1
        for (int I = 5; I < I_norm; ++I) interval(I, K0{});
        interval(I_norm, std::integral_constant<int, 3>{});
        interval(I_norm + 1, K0{});
        interval(I_norm + 2, std::integral_constant<int, 4>{});
#pragma unroll 1
        for (int I = I_norm + 3; I < nst - 1; ++I) interval(I, K0{});
        interval(nst - 1, std::integral_constant<int, 7>{});

        VQP_STAMP_END(4);
        {   // both row blocks of the last tile
            const float o10 = m1[0], o20 = m2[0], o11 = m1[1], o21 = m2[1];
#pragma unroll
            for (int e = 0; e < 16; ++e) { fold1(accB[0], e, 0); fold1(accB[1], e, 1); }
            book(0, o10, o20, ptile);
            book(1, o11, o21, ptile);
        }

        VQP_STAMP_END(5);
        // ---- merge the half-waves, classify (certified / pair / open), emit the index; lane l then carries row l of the 64 ----
        int code, cls, id2;
        {
            int codes[2], id2s[2];
            bool cert[2], pairf[2];
            float dbg4[2][3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int e1 = (int)(__float_as_uint(m1[t]) & 15u), e2b = (int)(__float_as_uint(m2[t]) & 15u);
                const int ia1 = tix[t] * 32 + 8 * (e1 >> 2) + 4 * half + (e1 & 3);
                const int ia2 = tix2[t] * 32 + 8 * (e2b >> 2) + 4 * half + (e2b & 3);
                float b1, b2, b3;
                merge3(m1[t], m2[t], m3[t], ia1, ia2, xor32f(m1[t]), xor32f(m2[t]), xor32f(m3[t]), (int)xor32((unsigned)ia1), (int)xor32((unsigned)ia2),
                       b1, b2, b3, codes[t], id2s[t]);
                const float thr = eps_c[t] * SS_c + 8e-6f * fabsf(b1);
                cert[t] = ((b1 - b2) > thr) && codes[t] < a.C;
                pairf[t] = !cert[t] && ((b1 - b3) > thr) && codes[t] < a.C && id2s[t] < a.C;
                dbg4[t][0] = b1 * iSS_c; dbg4[t][1] = b2 * iSS_c; dbg4[t][2] = thr * iSS_c;
            }
            code = half ? codes[1] : codes[0];
            id2 = half ? id2s[1] : id2s[0];
            const bool certified = half ? cert[1] : cert[0];
            const bool pair = half ? pairf[1] : pairf[0];
            const int64_t row = (int64_t)sb * 256 + wave * 64 + lane;
            const bool on = row < a.N;
            if (on) {
                a.idx_out[row * a.idx_stride] = (int64_t)(code < a.C ? code : 0);
                if (a.dbg) {
                    float *d = a.dbg + row * 4;
                    d[0] = half ? dbg4[1][0] : dbg4[0][0]; d[1] = half ? dbg4[1][1] : dbg4[0][1]; d[2] = half ? dbg4[1][2] : dbg4[0][2];
                    d[3] = certified ? 0.f : (pair ? 2.f : 1.f);
                }
            }
            if (code >= a.C) code = 0;
            cls = !on ? 0 : (certified ? 0 : (pair ? 2 : 1));
            prow = (int)row;
        }
        pbalo = __ballot(cls == 1);
        pbalp = __ballot(cls == 2);
        // list space for this piece's open / pair rows: ONE 64-bit atomic on (flag_count[0], flag_count[1]) by lane 0, and only if there
        // is something to append.  As an asm statement: hipcc would wait for the returned value at the end of the `if (lane == 0)`
        // it needs (a round trip to L2 with the SIMD idle); this way the rest of the block's epilogue runs under it, and the wait
        // (below, naming the destination) is the last thing before the next block starts.
        unsigned long long pbase64;
        VQP_STAMP_END(6);
        {
#ifndef VQP_NO_ATOMIC
            const int em = __builtin_amdgcn_readfirstlane((pbalo | pbalp) ? 1 : 0);   // exec for the atomic: lane 0 or nobody
#else
            const int em = 0;
#endif
            const unsigned long long cnt = (unsigned long long)__popcll(pbalo) | ((unsigned long long)__popcll(pbalp) << 32);
            unsigned long long keep;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %5\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add_x2 %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                         : "=&v"(pbase64), "=&s"(keep) : "v"(0u), "v"(cnt), "s"(a.flag_count), "s"(em) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        pcode = code; pcls = cls; pid2 = id2;
        {
            const int64_t prow0 = (int64_t)sb * 256 + wave * 64;
            const int64_t lim = a.N - 1 - prow0;
            plim = lim > 63 ? 63 : (lim < 0 ? 0 : (int)lim);
            pqbase = (char *)a.q_out + (prow0 < a.N ? prow0 : a.N - 1) * a.ldq * 2;
        }
        has_prev = true;
        // ---- the next piece becomes the current one (its operands were converted during the last tile) ----
        SS_c = __uint_as_float((unsigned)(SX_n + sc + 127) << 23);
        iSS_c = __uint_as_float((unsigned)(127 - SX_n - sc) << 23);
        eps_c[0] = eps_of(xs2_n[0], SX_n);
        eps_c[1] = eps_of(xs2_n[1], SX_n);
        reset_fold();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pbase64) :: "memory");
#ifdef VQP_NO_ATOMIC
        pbase64 = 0ull;
#endif
        pbase_o = __builtin_amdgcn_readfirstlane((int)(unsigned)pbase64);
        pbase_p = __builtin_amdgcn_readfirstlane((int)(unsigned)(pbase64 >> 32));
        VQP_STAMP_END(7);
        if (!has_next) break;
    }

    // ---- tail: the last piece's q rows and list entries ----
    if (HASQ) {
#pragma unroll 4
        for (int t = 0; t < 32; ++t) {
            const int c = __builtin_amdgcn_ds_bpermute((2 * t + half) * 4, pcode);
            const uint4 g = *(const uint4 *)((const char *)a.embed_bf16 + ((unsigned)c * (unsigned)(DT * 2) + j16));
            const int rl = min(2 * t + half, plim);
            *(uint4 *)(pqbase + ((unsigned)(rl * ldq2) + j16)) = g;
        }
    }
    write_lists(pbalo, pbalp, pbase_o, pbase_p, pcls, prow, pcode, pid2);
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
#ifndef VQP_MIN_ROWS
#define VQP_MIN_ROWS (256 * 256)      // below one super-block per CU the 4-wave kernel spreads the rows better
#endif

static long long *vqp_g_trace = nullptr;              // dev builds with -DVQP_TRACE: where the kernel puts its s_memtime stamps
extern "C" void vqhip_screenp_set_trace(long long *p) { vqp_g_trace = p; }

// Opt-in while it is being tuned (VQHIP_SCREEN_PERSIST=1).
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
    if (a.n_tiles16 < VQP_MIN_TILES || (a.n_tiles16 & 3) || a.n_tiles16 * 32 > VQP_MAX_NORM_CODES) return 0;
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
    hipLaunchKernelGGL((vq_screenp_kernel<256, METRIC, HASQ>), dim3((unsigned)grid), dim3(256), smem, st, a, nsb, vqp_g_trace);
    return vq_launch_status("vq_screenp_kernel");
}

int vq_screenp_launch(const ScreenArgs &a, int metric_is_cosine, hipStream_t st)
{
    if (a.q_out) return metric_is_cosine ? vqp_launch<1, true>(a, st) : vqp_launch<0, true>(a, st);
    return metric_is_cosine ? vqp_launch<1, false>(a, st) : vqp_launch<0, false>(a, st);
}
