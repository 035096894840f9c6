// vq_wide.hip -- codebook dims 512 < D <= 2048, gfx950 only.
//
// The reference takes any `dim` (vqp.py:803-806; cdist :58-62 is dimension-agnostic).  The tuned kernels of vqhip.hip / vq_screen.hip keep
// a wave's rows resident in VGPRs as MFMA operands, which ends at D = 512 (256 registers); none of BASELINE's configurations or the
// README's examples goes beyond that, so the wide dims get a plain, exact path instead of a fast one:
//
//   vq_wide_pack_kernel     ||c||^2 per code in ATen's CPU summation order (with the cascade level that engages beyond 512 elements,
//                           oracle/vq_oracle.c::aten_sumsq_row) + the bf16 copy of the codebook (q rows of bf16 I/O)
//   vq_wide_assign_kernel   nearest code per row on the VALU: 64 rows x 64 codes per step, a 4 x 4 micro-tile per thread, every
//                           x . c ONE ascending fp32 FMA chain (the numerics contract of DESIGN 2: what v_mfma_f32_32x32x2_f32 fed
//                           k = 0, 1, 2, ... computes, what oracle/vq_oracle.c::vqo_assign defines), then exactly
//                           (x2 + y2) + (-2 xy), max(., 1e-8), correctly rounded sqrt, first minimum -- indices, winning distance, q rows,
//                           squared-error partials as vq_assign_kernel produces them
//   vq_wide_row_sumsq / vq_wide_l2norm_kernel   the row norms of the above as entry points of their own
//   vq_wide_embed_kernel    ema_inplace of embed_avg + update_ema's division (vqp.py:76-97, 576-584) for rows wider than 512
//   vq_wide_decode_kernel   codebook[indices] summed over the quantizers (vqp.py:1003, rvq.py:341-381)
//
// Everything else a wide module needs (statistics: vq_segsum_fast_kernel; routing: vq_route_kernel<.., 16 / 32, 64>) is the general
// form of the kernels in vqhip.hip.  Options that read whole score rows (top-k, cross-entropy, diversity, gumbel), the screened
// search and the fused residual loop are not available beyond D = 512 (they raise).

#include <math.h>
#include <stdlib.h>

#include "vqhip_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

int vq_launch_status(const char *what);

__device__ __forceinline__ float wd_bf16_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short wd_f32_to_bf16(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float wd_round_bf16(float f) { return wd_bf16_to_f32(wd_f32_to_bf16(f)); }

template <bool BF16>
__device__ __forceinline__ float wd_load(const void *base, int64_t off)
{
    if (BF16) return wd_bf16_to_f32(((const unsigned short *)base)[off]);
    return ((const float *)base)[off];
}

// ATen's CPU order of sum(x * x) over one contiguous row (vqp.py:59-60), any D <= 2048: 8 SIMD lanes x 4 interleaved accumulators
// = 32 chains; every 16 "rows" of 32 elements the chains are folded into a second level and restart from zero (the cascade of
// ATen's vectorized sum: level_power = max(4, ceil_log2(D / 32) / 4) = 4 for D <= 2048, and the third level never engages below
// 8192 elements); then leftover vectors, ((a0 + a1) + a2) + a3 over the accumulators, the scalar tail, the 8 lanes left to right.
// Restates oracle/vq_oracle.c::aten_sumsq_row (tests/test_oracle.py pins that against torch for D up to 2048).
template <typename F>
__device__ float wd_aten_sumsq(F ld, int D)
{
    const int V = D >> 3;
    const int size = V >> 2;
    float a0[32], a1[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { a0[c] = 0.f; a1[c] = 0.f; }
    int i = 0;
    for (; i + 16 <= size;) {
        for (int j = 0; j < 16; ++j, ++i) {
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const float v = ld(32 * i + c);
                a0[c] += v * v;
            }
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) { a1[c] += a0[c]; a0[c] = 0.f; }
    }
    for (; i < size; ++i) {
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const float v = ld(32 * i + c);
            a0[c] += v * v;
        }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) a0[c] += a1[c];
    for (int v = size * 4; v < V; ++v) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const float t = ld(v * 8 + l);
            a0[l] += t * t;
        }
    }
    float fin = 0.f;
    for (int e = V * 8; e < D; ++e) {
        const float t = ld(e);
        fin += t * t;
    }
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float p = ((a0[l] + a0[8 + l]) + a0[16 + l]) + a0[24 + l];
        fin += p;
    }
    return fin;
}

// ---- packed layout for wide dims: y2 [C] floats (256-byte padded) | bf16 copy [C, D] ------------------------------------------
// The same sum by 32 cooperating lanes (a half-wave; c = the lane's chain, 0 .. 31): chain c of wd_aten_sumsq is exactly the elements
// c, c + 32, c + 64, ... with the cascade fold every 16 of them, so the 32 lanes read 128 contiguous bytes per step instead of one
// thread walking a whole row; the folds, the leftover vectors (chains 0 .. 7), the scalar tail and the final left-to-right sum keep
// their order.  Every lane of the half-wave returns the row's sum.  `ld` as above.
template <typename F>
__device__ __forceinline__ float wd_aten_sumsq_coop(F ld, int D, int c)
{
    const int V = D >> 3;
    const int size = V >> 2;
    float a0 = 0.f, a1 = 0.f;
    int i = 0;
    // (the 16 loads of a block are requested together and consumed in order: as a load-then-add loop every step of the chain waited for
    //  its own load -- 24 .. 64 round trips to L2 / HBM per row, eight rows at a time, which was a third of vq_wide_mfma_kernel's time)
    for (; i + 16 <= size; i += 16) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ld(32 * (i + j) + c);
#pragma unroll
        for (int j = 0; j < 16; ++j) a0 += v[j] * v[j];
        a1 += a0; a0 = 0.f;
    }
    {
        float v[16];
#pragma unroll
        for (int j = 0; j < 15; ++j) v[j] = (i + j < size) ? ld(32 * (i + j) + c) : 0.f;
#pragma unroll
        for (int j = 0; j < 15; ++j)
            if (i + j < size) a0 += v[j] * v[j];
    }
    a0 += a1;
    for (int v = size * 4; v < V; ++v)
        if (c < 8) {
            const float t = ld(v * 8 + c);
            a0 += t * t;
        }
    float fin = 0.f;
    for (int e = V * 8; e < D; ++e) {
        const float t = ld(e);
        fin += t * t;
    }
    const int base = (int)(threadIdx.x & 32);          // first lane of this half-wave inside its wave
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float p = ((__shfl(a0, base + l, 64) + __shfl(a0, base + 8 + l, 64)) + __shfl(a0, base + 16 + l, 64)) + __shfl(a0, base + 24 + l, 64);
        fin += p;
    }
    return fin;
}

static inline size_t wd_align(size_t v, size_t a) { return (v + a - 1) / a * a; }
size_t vq_wide_packed_bytes(int C, int D) { return wd_align((size_t)C * 4, 256) + wd_align((size_t)C * D * 2, 256) + 256; }
static inline size_t wd_bf16_offset(int C) { return wd_align((size_t)C * 4, 256); }

// A half-wave per code for the norm (eight codes per workgroup: C / 8 workgroups; one thread per code walked a 3 - 8 KiB row by itself
// from 1 - 4 workgroups -- 190 - 310 us of a 0.9 - 1.6 ms step, round 6), the bf16 copy grid-strided over the same workgroups.
__global__ void __launch_bounds__(256) vq_wide_pack_kernel(const float *embed, int C, int D, float *y2, unsigned short *ebf, size_t in_hs, size_t out_hs)
{
    embed += blockIdx.y * in_hs;
    y2 = (float *)((char *)y2 + blockIdx.y * out_hs);
    ebf = (unsigned short *)((char *)ebf + blockIdx.y * out_hs);
    const int c = blockIdx.x * 8 + (int)(threadIdx.x >> 5);
    const float *row = embed + (size_t)(c < C ? c : C - 1) * D;
    const float v = wd_aten_sumsq_coop([&](int e) { return row[e]; }, D, (int)(threadIdx.x & 31));
    if (c < C && (threadIdx.x & 31) == 0) y2[c] = v;
    const size_t n = (size_t)C * D;
    if ((D & 3) == 0 && (((uintptr_t)embed) & 15) == 0 && (((uintptr_t)ebf) & 7) == 0) {
        for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (size_t)gridDim.x * 1024) {
            const f32x4 w = *(const f32x4 *)(embed + e);
            uint2 o;
            o.x = (unsigned)wd_f32_to_bf16(w.x) | ((unsigned)wd_f32_to_bf16(w.y) << 16);
            o.y = (unsigned)wd_f32_to_bf16(w.z) | ((unsigned)wd_f32_to_bf16(w.w) << 16);
            *(uint2 *)(ebf + e) = o;
        }
    } else {
        for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) ebf[e] = wd_f32_to_bf16(embed[e]);
    }
}

int vq_wide_pack(const float *embed, int C, int D, float *packed, int H, void *stream)
{
    if (((uintptr_t)packed) & 15) VQ_FAIL(VQHIP_EALIGN, "pack_codebook: packed must be 16-byte aligned");
    char *base = (char *)packed;
    hipLaunchKernelGGL(vq_wide_pack_kernel, dim3((unsigned)((C + 7) / 8), H), dim3(256), 0, (hipStream_t)stream, embed, C, D, (float *)base,
                       (unsigned short *)(base + wd_bf16_offset(C)), (size_t)C * D, vq_wide_packed_bytes(C, D));
    return vq_launch_status("vq_wide_pack_kernel");
}

// ---- row norms ------------------------------------------------------------------------------------------------------------------
template <bool XBF16>
__global__ void __launch_bounds__(256) vq_wide_sumsq_kernel(const void *x, int64_t N, int D, int64_t ldx, float *out)
{
    // a half-wave per row (eight rows per workgroup): the 32 chains of ATen's order on 32 lanes, 128 contiguous bytes per step
    const int64_t n0 = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int64_t n = n0 < N ? n0 : N - 1;
    const float v = wd_aten_sumsq_coop([&](int e) { return wd_load<XBF16>(x, n * ldx + e); }, D, (int)(threadIdx.x & 31));
    if (n0 < N && (threadIdx.x & 31) == 0) out[n0] = v;
}

int vq_wide_row_sumsq(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, float *out, void *stream)
{
    if (N <= 0) return 0;
    const unsigned blocks = (unsigned)((N + 7) / 8);
    if (x_dtype == VQHIP_BF16) hipLaunchKernelGGL(vq_wide_sumsq_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, D, ldx, out);
    else hipLaunchKernelGGL(vq_wide_sumsq_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, D, ldx, out);
    return vq_launch_status("vq_wide_sumsq_kernel");
}

// l2norm of rows (vqp.py:37-38 at :1159): out = x / max(||x||, 1e-6), ||x||^2 in ATen order; bf16 tensors: norm and quotient rounded
// to bf16 like the reference's bf16 ops -- the arithmetic of vqhip_l2norm_rows.  One wave per row; both half-waves form the sum
// cooperatively (the same value: ATen's order), instead of lane 0 walking the row alone.
template <bool XBF16>
__global__ void __launch_bounds__(256) vq_wide_l2norm_kernel(const void *x, int64_t N, int D, int64_t ldx, void *out, int64_t ldo)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;         // (wave-uniform)
    const float x2 = wd_aten_sumsq_coop([&](int e) { return wd_load<XBF16>(x, n * ldx + e); }, D, lane & 31);
    float nrm = sqrtf(x2);
    if (XBF16) nrm = wd_round_bf16(nrm);
    nrm = fmaxf(nrm, XBF16 ? wd_round_bf16(1e-6f) : 1e-6f);
    for (int d = lane; d < D; d += 64) {
        const float v = wd_load<XBF16>(x, n * ldx + d) / nrm;
        if (XBF16) ((unsigned short *)out)[n * ldo + d] = wd_f32_to_bf16(v);
        else ((float *)out)[n * ldo + d] = v;
    }
}

int vq_wide_l2norm_rows(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, void *out, int64_t ldo, void *stream)
{
    const unsigned blocks = (unsigned)((N + 3) / 4);
    if (x_dtype == VQHIP_BF16) hipLaunchKernelGGL(vq_wide_l2norm_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, D, ldx, out, ldo);
    else hipLaunchKernelGGL(vq_wide_l2norm_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, D, ldx, out, ldo);
    return vq_launch_status("vq_wide_l2norm_kernel");
}

// ---- assignment -------------------------------------------------------------------------------------------------------------------
struct WideAssignArgs {
    const void *x;
    int64_t N;
    int D;
    int64_t ldx;
    const float *y2;            // [C] ||c||^2, ATen order
    const float *embed;         // [C, D] fp32
    int C;
    int64_t *idx_out;
    void *q_out;
    int q_bf16;
    int64_t ldq;
    float *best_out;
    float *rnorm_out;
    double *sqerr_partial;      // one entry per workgroup (128 rows)
    const uint8_t *row_mask;
    int skip_norm;
    int vec;                    // D % 8 == 0 and x / embed rows 16-byte aligned: the MFMA kernel's staging uses 16-byte loads
};

#define WD_ROWS 64
#define WD_CODES 64
#define WD_KC 16
#define WD_PITCH 68

// METRIC 0: Euclidean; 1: cosine (rows l2-normalised here unless skip_norm)
template <bool XBF16, int METRIC>
__global__ void __launch_bounds__(256) vq_wide_assign_kernel(const WideAssignArgs a)
{
    __shared__ float xs[WD_KC][WD_PITCH], cs[WD_KC][WD_PITCH];
    __shared__ float s_x2[WD_ROWS], s_nrm[WD_ROWS];
    __shared__ float s_bd[WD_ROWS][17];
    __shared__ int s_bi[WD_ROWS][17];
    __shared__ int s_win[WD_ROWS];
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = tid >> 4, tx = tid & 15;          // micro-tile: rows 4 ty .. + 3, codes 4 tx .. + 3 of the 64 x 64 step
    const int lr = tid >> 2, lk = (tid & 3) * 4;     // tile loads: row / code lr, elements lk .. lk + 3 of the 16-element k chunk
    double ds = 0.0;
    for (int sub = 0; sub < VQHIP_ASSIGN_ROWS_PER_BLOCK / WD_ROWS; ++sub) {
        const int64_t r0 = (int64_t)blockIdx.x * VQHIP_ASSIGN_ROWS_PER_BLOCK + sub * WD_ROWS;
        if (r0 >= a.N) break;
        __syncthreads();
        // ---- row norms: ATen order, one thread per row (sequential by definition) ----
        if (tid < WD_ROWS) {
            const int64_t r = r0 + tid < a.N ? r0 + tid : a.N - 1;
            const float x2 = wd_aten_sumsq([&](int e) { return wd_load<XBF16>(a.x, r * a.ldx + e); }, a.D);
            float nrm = 1.f;
            if (METRIC == 1 && !a.skip_norm) {       // l2norm (vqp.py:37-38): bf16 tensors normalise in bf16
                nrm = sqrtf(x2);
                if (XBF16) nrm = wd_round_bf16(nrm);
                nrm = fmaxf(nrm, XBF16 ? wd_round_bf16(1e-6f) : 1e-6f);
            }
            s_x2[tid] = x2;
            s_nrm[tid] = nrm;
        }
        __syncthreads();
        float bd[4], x2r[4];
        int bi[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { bd[i] = (METRIC == 0) ? INFINITY : -INFINITY; bi[i] = 0; x2r[i] = s_x2[4 * ty + i]; }
        const int64_t xrow = r0 + lr < a.N ? r0 + lr : a.N - 1;
        const float xnrm = s_nrm[lr];
        for (int c0 = 0; c0 < a.C; c0 += WD_CODES) {
            float acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
            const int crow = c0 + lr < a.C ? c0 + lr : a.C - 1;
            for (int k0 = 0; k0 < a.D; k0 += WD_KC) {
                float xv[4], cv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + lk + e;
                    float v = (k < a.D) ? wd_load<XBF16>(a.x, xrow * a.ldx + k) : 0.f;
                    if (METRIC == 1 && !a.skip_norm) {
                        v = v / xnrm;
                        if (XBF16) v = wd_round_bf16(v);
                    }
                    xv[e] = v;
                    cv[e] = (k < a.D) ? a.embed[(size_t)crow * a.D + k] : 0.f;
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < 4; ++e) { xs[lk + e][lr] = xv[e]; cs[lk + e][lr] = cv[e]; }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < WD_KC; ++k) {
                    const f32x4 xq = *(const f32x4 *)&xs[k][4 * ty];
                    const f32x4 cq = *(const f32x4 *)&cs[k][4 * tx];
                    const float xa[4] = {xq.x, xq.y, xq.z, xq.w}, ca[4] = {cq.x, cq.y, cq.z, cq.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(xa[i], ca[j], acc[i][j]);   // ONE ascending chain per (row, code)
                }
            }
            // ---- this step's 4 codes per row, ascending: strict comparison keeps the first extremum ----
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int code = c0 + 4 * tx + j;
                if (code < a.C) {
                    const float y2 = (METRIC == 0) ? a.y2[code] : 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (METRIC == 0) {
                            const float t = x2r[i] + y2;
                            const float d = sqrtf(fmaxf(__builtin_fmaf(-2.0f, acc[i][j], t), 1e-8f));
                            if (d < bd[i]) { bd[i] = d; bi[i] = code; }
                        } else {
                            if (acc[i][j] > bd[i]) { bd[i] = acc[i][j]; bi[i] = code; }
                        }
                    }
                }
            }
        }
        // ---- a row's 16 threads (tx = 0 .. 15) meet: best score, then lowest index ----
#pragma unroll
        for (int i = 0; i < 4; ++i) { s_bd[4 * ty + i][tx] = bd[i]; s_bi[4 * ty + i][tx] = bi[i]; }
        __syncthreads();
        if (tid < WD_ROWS) {
            float b = s_bd[tid][0];
            int w = s_bi[tid][0];
            for (int t = 1; t < 16; ++t) {
                const float o = s_bd[tid][t];
                const int oi = s_bi[tid][t];
                const bool take = (METRIC == 0) ? ((o < b) || (o == b && oi < w)) : ((o > b) || (o == b && oi < w));
                b = take ? o : b;
                w = take ? oi : w;
            }
            s_win[tid] = w;
            const int64_t r = r0 + tid;
            if (r < a.N) {
                a.idx_out[r] = (int64_t)w;
                if (a.best_out) a.best_out[r] = b;
                if (a.rnorm_out) a.rnorm_out[r] = (METRIC == 0) ? s_x2[tid] : s_nrm[tid];
            }
        }
        __syncthreads();
        // ---- q rows and the commitment loss' squared error: one wave per row, 16 rows per wave ----
        if (a.q_out || a.sqerr_partial) {
            for (int rr = wave; rr < WD_ROWS; rr += 4) {
                const int64_t r = r0 + rr;
                if (r >= a.N) break;
                const int c = s_win[rr];
                const float nr = s_nrm[rr];
                const bool counted = !a.row_mask || a.row_mask[r] != 0;
                float ls = 0.f;
                for (int d = lane; d < a.D; d += 64) {
                    float g = a.embed[(size_t)c * a.D + d];
                    if (a.q_bf16) g = wd_round_bf16(g);
                    if (a.q_out) {
                        if (a.q_bf16) ((unsigned short *)a.q_out)[r * a.ldq + d] = wd_f32_to_bf16(g);
                        else ((float *)a.q_out)[r * a.ldq + d] = g;
                    }
                    if (a.sqerr_partial) {
                        float xv = wd_load<XBF16>(a.x, r * a.ldx + d);
                        if (METRIC == 1 && !a.skip_norm) {
                            xv = xv / nr;
                            if (XBF16) xv = wd_round_bf16(xv);
                        }
                        const float df = g - xv;
                        ls += df * df;
                    }
                }
                if (a.sqerr_partial && counted) ds += (double)ls;
            }
        }
    }
    if (a.sqerr_partial) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ds += __shfl_xor(ds, o, 64);
        __syncthreads();
        if (lane == 0) s_red[wave] = ds;
        __syncthreads();
        if (tid == 0) a.sqerr_partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

// ---- the same search on the fp32 MFMA pipe (round 6) ---------------------------------------------------------------------------------
// vq_wide_assign_kernel above runs every dot product as an FMA chain on the VALU: 31 algorithmic TFLOP/s.  v_mfma_f32_32x32x2_f32 fed the
// features in ascending order computes the SAME chain (the numerics contract of the tuned exact kernel; two features per
// instruction: lanes 0..31 carry feature 2p, lanes 32..63 feature 2p + 1, accumulated in that order), so beyond D = 512 only the operand
// plumbing changes: the rows no longer fit a wave's registers, hence
//   * a workgroup = 128 rows (4 waves x 32 rows, the MFMA's B operand) x WM_T code tiles of 32 codes (A operand) at a time;
//   * the feature axis in slabs of WM_KS = 64: the slab of the 128 rows and of the WM_T x 32 codes is staged through LDS in the MFMAs'
//     operand order (global loads of the NEXT slab are in flight in registers while this one is multiplied), a wave keeps its B slab
//     in 32 registers for all WM_T tiles, and the accumulators run on over the slabs -- ascending k, one chain per (row, code);
//   * after the last slab: (x2 + y2) + (-2 xy), max(., 1e-8), correctly rounded sqrt, strict < in ascending code order per lane, the two
//     half-waves merged by (distance, index) -- the arithmetic of the kernel above, bit for bit (tests/test_gpu_ops.py: indices and
//     winning distances equal to the chain oracle at D = 640 .. 2048).
// Rows are re-staged once per group of WM_T tiles (from L2: C / (32 WM_T) times 128 x D x 4 bytes per workgroup).
#define WM_T 4
#ifndef WM_KS
#define WM_KS 32            // 32: 68 KiB of LDS, two workgroups per CU (0.64 of the nominal pipe rate at dim 768 x 65 536 rows); 64: one (0.57)
#endif
#define WM_ITS (WM_KS / 16)                            // staging work items per thread and operand
#define WM_G8 (WM_KS / 8)                              // 8-feature pieces (= pair groups) of a slab
typedef float f32x16w __attribute__((ext_vector_type(16)));

// A unit = the 32 lanes' 16-byte pieces of one (tile or wave, pair group, hi): 512 bytes + 16 bytes of padding, so that the staging
// writes -- 16 lanes = 2 rows x 8 pair groups per LDS pass, a pair group every 2 units -- land 32 bytes apart per pair group and 16 per
// row: all 64 banks.  (Unpadded, the 8 pair groups of a row hit the same banks: every staging write an 8-way conflict, ~4 000 LDS
// cycles per slab and workgroup -- what held the kernel at 0.55 - 0.6 of the pipe through every other change.)
#define WM_UNIT (128 + 32 / WM_G8)      // 132 at WM_KS = 64 (pair groups 32 bytes apart), 136 at WM_KS = 32 (64 bytes apart)
#define WM_IMG (WM_T * (WM_KS / 8) * 2 * WM_UNIT)      // floats of one operand image (33 KiB)
#define WM_SMEM (4 * WM_IMG * 4)                       // bytes: A and B images, two buffers each (128 KiB: one workgroup per CU)
template <bool XBF16, int METRIC>
__global__ void __launch_bounds__(256, WM_KS <= 32 ? 2 : 1) vq_wide_mfma_kernel(const WideAssignArgs a)
{
    // operand images: [tile or wave][p4 = pair group of 4][hi][lane 0..31][4 pairs] floats -- a lane's ds_read_b128 returns the values of
    // four consecutive MFMAs; consecutive lanes read consecutive 16-byte pieces (conflict-free).  TWO buffers per operand: the next
    // slab is written into the other one from inside this slab's MFMA stream, one barrier per slab (the single-buffered first version
    // -- barrier, write, barrier, multiply, two workgroups per CU to cover for each other -- ran at 0.66 of the pipe's rate at its
    // actual clock; the exact kernel of the tuned dims, which stages inside its MFMA stream, at 0.95)
    extern __shared__ __attribute__((aligned(16))) float wm_smem[];
    float *const sA0 = wm_smem, *const sB0 = wm_smem + 2 * WM_IMG;
    __shared__ float s_x2[VQHIP_ASSIGN_ROWS_PER_BLOCK], s_nrm[VQHIP_ASSIGN_ROWS_PER_BLOCK];
    __shared__ int s_win[VQHIP_ASSIGN_ROWS_PER_BLOCK];
    __shared__ double s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * VQHIP_ASSIGN_ROWS_PER_BLOCK;
    const int n_slab = (a.D + WM_KS - 1) / WM_KS;
    const int n_grp = (a.C + 32 * WM_T - 1) / (32 * WM_T);

    // ---- row norms: ATen order, 32 lanes per row (8 rows per pass of the workgroup) ----
    for (int rr = tid >> 5; rr < VQHIP_ASSIGN_ROWS_PER_BLOCK; rr += 8) {
        const int64_t r = r0 + rr < a.N ? r0 + rr : a.N - 1;
        const float x2 = wd_aten_sumsq_coop([&](int e) { return wd_load<XBF16>(a.x, r * a.ldx + e); }, a.D, tid & 31);
        float nrm = 1.f;
        if (METRIC == 1 && !a.skip_norm) {       // l2norm (vqp.py:37-38): bf16 tensors normalise in bf16
            nrm = sqrtf(x2);
            if (XBF16) nrm = wd_round_bf16(nrm);
            nrm = fmaxf(nrm, XBF16 ? wd_round_bf16(1e-6f) : 1e-6f);
        }
        if ((tid & 31) == 0) { s_x2[rr] = x2; s_nrm[rr] = nrm; }
    }
    __syncthreads();

    // staging: work item (r, g8) = 8 consecutive features g8 of row / code r of the slab -> two 16-byte LDS pieces (hi = 0: the even
    // features, hi = 1: the odd ones); 128 x 8 items each for A and B, four per thread and operand; g8 runs fastest over the threads
    // (eight threads read 256 contiguous bytes of one row)
    // (vector path: a wave-uniform base per slab / group + a 32-bit lane offset fixed for the whole launch -- the address arithmetic of
    //  the staging was ~200 VALU instructions per slab and wave, and on this chip VALU work is taken out of the fp32 MFMA stream)
    constexpr int ES = XBF16 ? 2 : 4;
    unsigned xoff[WM_ITS], coff[WM_ITS];
#pragma unroll
    for (int it = 0; it < WM_ITS; ++it) {
        const int item = it * 256 + tid, r = item / WM_G8, g8 = item % WM_G8;
        const int64_t lim = a.N - 1 - r0;                                    // rows past the end repeat the last one
        const int rl = (int64_t)r < lim ? r : (int)(lim < 0 ? 0 : lim);
        xoff[it] = (unsigned)((int64_t)rl * a.ldx + g8 * 8) * (unsigned)ES;
        coff[it] = (unsigned)(r * a.D + g8 * 8) * 4u;
    }
    const char *const xbase = (const char *)a.x + r0 * a.ldx * ES;
    const bool fast = a.vec && !(METRIC == 1 && !a.skip_norm) && a.ldx < ((int64_t)1 << 22);      // (the lane offsets stay below 2^32 bytes)
    auto fetch = [&](int slab, int grp, float (&ra)[WM_ITS][8], float (&rb)[WM_ITS][8]) __attribute__((always_inline)) {
        if (fast) {
            const int k0 = slab * WM_KS;
            const char *const xs = xbase + (size_t)k0 * ES;
            const int c0 = grp * 32 * WM_T;
            const bool whole = k0 + WM_KS <= a.D && c0 + 32 * WM_T <= a.C;      // wave-uniform: nothing of this slab / group is padding
            const char *const cs = (const char *)(a.embed + (size_t)c0 * a.D + k0);
#pragma unroll
            for (int it = 0; it < WM_ITS; ++it) {
                const int item = it * 256 + tid, r = item / WM_G8, g8 = item % WM_G8;
                bool in = true, cin = true;
                if (!whole) { in = k0 + g8 * 8 < a.D; cin = in && c0 + r < a.C; }
                if (XBF16) {
                    const uint4 w = in ? *(const uint4 *)(xs + xoff[it]) : make_uint4(0u, 0u, 0u, 0u);
                    const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rb[it][2 * e] = __uint_as_float(ww[e] << 16); rb[it][2 * e + 1] = __uint_as_float(ww[e] & 0xffff0000u); }
                } else {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 w0 = in ? *(const f32x4 *)(xs + xoff[it]) : z, w1 = in ? *(const f32x4 *)(xs + xoff[it] + 16) : z;
                    rb[it][0] = w0.x; rb[it][1] = w0.y; rb[it][2] = w0.z; rb[it][3] = w0.w; rb[it][4] = w1.x; rb[it][5] = w1.y; rb[it][6] = w1.z; rb[it][7] = w1.w;
                }
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 q0 = cin ? *(const f32x4 *)(cs + coff[it]) : z, q1 = cin ? *(const f32x4 *)(cs + coff[it] + 16) : z;
                ra[it][0] = q0.x; ra[it][1] = q0.y; ra[it][2] = q0.z; ra[it][3] = q0.w; ra[it][4] = q1.x; ra[it][5] = q1.y; ra[it][6] = q1.z; ra[it][7] = q1.w;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < WM_ITS; ++it) {
            const int item = it * 256 + tid, r = item / WM_G8, g8 = item % WM_G8, k = slab * WM_KS + g8 * 8;
            const int64_t xr = r0 + r < a.N ? r0 + r : a.N - 1;
            const int code = grp * 32 * WM_T + r;
            const float inv = s_nrm[r];
            if (a.vec) {        // (D % 8 == 0: a piece is inside the row or past its end as a whole)
                const bool in = k < a.D;
                if (XBF16) {
                    const uint4 w = in ? *(const uint4 *)((const unsigned short *)a.x + xr * a.ldx + k) : make_uint4(0u, 0u, 0u, 0u);
                    const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rb[it][2 * e] = __uint_as_float(ww[e] << 16); rb[it][2 * e + 1] = __uint_as_float(ww[e] & 0xffff0000u); }
                } else {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 w0 = in ? *(const f32x4 *)((const float *)a.x + xr * a.ldx + k) : z, w1 = in ? *(const f32x4 *)((const float *)a.x + xr * a.ldx + k + 4) : z;
                    rb[it][0] = w0.x; rb[it][1] = w0.y; rb[it][2] = w0.z; rb[it][3] = w0.w; rb[it][4] = w1.x; rb[it][5] = w1.y; rb[it][6] = w1.z; rb[it][7] = w1.w;
                }
                const bool cin = in && code < a.C;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 c0 = cin ? *(const f32x4 *)(a.embed + (size_t)code * a.D + k) : z, c1 = cin ? *(const f32x4 *)(a.embed + (size_t)code * a.D + k + 4) : z;
                ra[it][0] = c0.x; ra[it][1] = c0.y; ra[it][2] = c0.z; ra[it][3] = c0.w; ra[it][4] = c1.x; ra[it][5] = c1.y; ra[it][6] = c1.z; ra[it][7] = c1.w;
                if (METRIC == 1 && !a.skip_norm) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { float v = rb[it][e] / inv; if (XBF16) v = wd_round_bf16(v); rb[it][e] = v; }
                }
                continue;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (k + e < a.D) ? wd_load<XBF16>(a.x, xr * a.ldx + k + e) : 0.f;
                if (METRIC == 1 && !a.skip_norm) {
                    v = v / inv;
                    if (XBF16) v = wd_round_bf16(v);
                }
                rb[it][e] = v;
                ra[it][e] = (code < a.C && k + e < a.D) ? a.embed[(size_t)code * a.D + k + e] : 0.f;
            }
        }
    };
    auto park_one = [&](int it, int buf, const float (&ra)[WM_ITS][8], const float (&rb)[WM_ITS][8]) __attribute__((always_inline)) {
        const int item = it * 256 + tid, r = item / WM_G8, g8 = item % WM_G8;
        // pairs 4 g8 .. 4 g8 + 3 of the slab: p4 = g8; image index ((blk * 8 + p4) * 2 + hi) * 32 + (r & 31), blk = r >> 5
        const int o = (((r >> 5) * (WM_KS / 8) + g8) * 2) * WM_UNIT + (r & 31) * 4 + buf * WM_IMG;
        float *pa = sA0 + o, *pb = sB0 + o;
        *(f32x4 *)pa = f32x4{ra[it][0], ra[it][2], ra[it][4], ra[it][6]};
        *(f32x4 *)(pa + WM_UNIT) = f32x4{ra[it][1], ra[it][3], ra[it][5], ra[it][7]};
        *(f32x4 *)pb = f32x4{rb[it][0], rb[it][2], rb[it][4], rb[it][6]};
        *(f32x4 *)(pb + WM_UNIT) = f32x4{rb[it][1], rb[it][3], rb[it][5], rb[it][7]};
    };

    float bd = (METRIC == 0) ? INFINITY : -INFINITY;
    float bs = INFINITY;            // Euclidean: the clamped squared distance whose root is bd
    int bi = 0;
    const float x2r = s_x2[wave * 32 + j];
    float ra[WM_ITS][8], rb[WM_ITS][8];
    fetch(0, 0, ra, rb);
#pragma unroll
    for (int it = 0; it < WM_ITS; ++it) park_one(it, 0, ra, rb);
    __syncthreads();
    int buf = 0;
    for (int grp = 0; grp < n_grp; ++grp) {
        f32x16w acc[WM_T];
#pragma unroll
        for (int t = 0; t < WM_T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int slab = 0; slab < n_slab; ++slab) {
            // the next slab (of this group, or the first one of the next group) is requested now, written into the other buffer from
            // the second half of this slab's MFMAs, and published by the barrier that ends the slab
            int ns = slab + 1, ng = grp;
            if (ns == n_slab) { ns = 0; ng = grp + 1; }
            const bool more = ng < n_grp;
#ifndef WM_DEV_NOFETCH          // dev builds (timing only, wrong results): the operands of the first slab are multiplied again and again
            if (more) fetch(ns, ng, ra, rb);
#endif
            const float *const sA = sA0 + buf * WM_IMG, *const sB = sB0 + buf * WM_IMG;
            f32x4 b[WM_KS / 8];
#pragma unroll
            for (int p4 = 0; p4 < WM_KS / 8; ++p4) b[p4] = *(const f32x4 *)(sB + ((wave * (WM_KS / 8) + p4) * 2 + hi) * WM_UNIT + j * 4);
            // One tile after the other, 32 MFMAs in a row on ONE accumulator (k ascending inside it, as before).  Taking turns between the
            // four accumulators after every MFMA -- the first version -- ran at 0.54 of the pipe's rate whatever was removed around
            // it (staging loads, epilogue, LDS prefetch distance: tools/time_wide_kernel.py with dev builds): an MFMA whose C operand is
            // not the previous MFMA's result moves its 16 accumulator registers through the register file, and at K = 2 that costs
            // about as much as the product itself; a dependent chain keeps them in the pipe (the exact kernel of the tuned dims has
            // always run one chain per wave).  A pieces: a ring of three, requested two steps ahead, pinned against the scheduler.
            constexpr int NQ = WM_T * (WM_KS / 8);
            f32x4 aq[3];
            aq[0] = *(const f32x4 *)(sA + (0 * 2 + hi) * WM_UNIT + j * 4);
            aq[1] = *(const f32x4 *)(sA + (1 * 2 + hi) * WM_UNIT + j * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int t = q / (WM_KS / 8), p4 = q % (WM_KS / 8);
                if (q + 2 < NQ) aq[(q + 2) % 3] = *(const f32x4 *)(sA + ((q + 2) * 2 + hi) * WM_UNIT + j * 4);
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[q % 3].x, b[p4].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[q % 3].y, b[p4].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[q % 3].z, b[p4].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[q % 3].w, b[p4].w, acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q >= NQ / 2 && (q - NQ / 2) % (NQ / 2 / WM_ITS) == 0) {       // WM_ITS times in the second half: a share of the next slab's operands
                    if (more) park_one((q - NQ / 2) / (NQ / 2 / WM_ITS), buf ^ 1, ra, rb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            buf ^= 1;
        }
        // ---- this group's codes, ascending per lane: register e of tile t <-> code 32 (grp WM_T + t) + 8 (e >> 2) + 4 hi + (e & 3) ----
        // Euclidean: the root is monotone, so a tile none of whose squared distances is below the best one's cannot hold a strictly
        // smaller distance for this lane -- one fma + min per score and a wave-level skip instead of a correctly rounded sqrt per score
        // (~25 VALU instructions each, taken out of the fp32 MFMA stream); a tile that may improve some lane runs the reference's
        // arithmetic score by score as before.
#pragma unroll
        for (int t = 0; t < WM_T; ++t) {
            const int cb = (grp * WM_T + t) * 32 + 4 * hi;
#ifdef WM_DEV_NOEPI             // dev builds (timing only, wrong results)
            if (acc[t][0] == 12345.f) bi = cb;
            continue;
#endif
            if (METRIC == 0) {
                float sv[16];
                float smin = INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int code = cb + 8 * (e >> 2) + (e & 3);
                    const float tt = x2r + a.y2[code < a.C ? code : a.C - 1];
                    sv[e] = code < a.C ? fmaxf(__builtin_fmaf(-2.0f, acc[t][e], tt), 1e-8f) : INFINITY;
                    smin = fminf(smin, sv[e]);
                }
                if (!__any(smin < bs)) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int code = cb + 8 * (e >> 2) + (e & 3);
                    if (sv[e] < bs) {                 // (sv >= bs: its root is >= bd, no strict improvement)
                        const float d = sqrtf(sv[e]);
                        if (d < bd) { bd = d; bi = code; bs = sv[e]; }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int code = cb + 8 * (e >> 2) + (e & 3);
                    if (code < a.C && acc[t][e] > bd) { bd = acc[t][e]; bi = code; }
                }
            }
        }
    }
    {   // the two half-waves of a row: better score, then lower index
        const float od = __shfl_xor(bd, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        const bool take = (METRIC == 0) ? ((od < bd) || (od == bd && oi < bi)) : ((od > bd) || (od == bd && oi < bi));
        bd = take ? od : bd;
        bi = take ? oi : bi;
    }
    const int lrow = wave * 32 + j;
    if (hi == 0) {
        s_win[lrow] = bi;
        const int64_t r = r0 + lrow;
        if (r < a.N) {
            a.idx_out[r] = (int64_t)bi;
            if (a.best_out) a.best_out[r] = bd;
            if (a.rnorm_out) a.rnorm_out[r] = (METRIC == 0) ? s_x2[lrow] : s_nrm[lrow];
        }
    }
    __syncthreads();
    // ---- q rows and the commitment loss' squared error: one wave per row (as vq_wide_assign_kernel) ----
    double ds = 0.0;
    // q rows alone (the training forward: the loss comes from the statistics pass), 16-byte accesses: four rows' pieces requested
    // together -- element by element, row by row every load waited for the store in front of it (no restrict): ~0.2 ms of a 1.25 ms step
    const bool q_fast = a.q_out && !a.sqerr_partial && (a.D & 3) == 0 && (((uintptr_t)a.embed) & 15) == 0 &&
                        (((uintptr_t)a.q_out) & (a.q_bf16 ? 7 : 15)) == 0 && ((a.ldq * (a.q_bf16 ? 2 : 4)) & (a.q_bf16 ? 7 : 15)) == 0;
    if (q_fast) {
        for (int rr = wave * 4; rr < VQHIP_ASSIGN_ROWS_PER_BLOCK; rr += 16) {
            for (int d = lane * 4; d < a.D; d += 256) {
                f32x4 g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = *(const f32x4 *)(a.embed + (size_t)s_win[rr + u] * a.D + d);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t r = r0 + rr + u;
                    if (r >= a.N) continue;
                    if (a.q_bf16) {
                        uint2 o;
                        o.x = (unsigned)wd_f32_to_bf16(g[u].x) | ((unsigned)wd_f32_to_bf16(g[u].y) << 16);
                        o.y = (unsigned)wd_f32_to_bf16(g[u].z) | ((unsigned)wd_f32_to_bf16(g[u].w) << 16);
                        *(uint2 *)((unsigned short *)a.q_out + r * a.ldq + d) = o;
                    } else {
                        *(f32x4 *)((float *)a.q_out + r * a.ldq + d) = g[u];
                    }
                }
            }
        }
    } else if (a.q_out || a.sqerr_partial) {
        for (int rr = wave; rr < VQHIP_ASSIGN_ROWS_PER_BLOCK; rr += 4) {
            const int64_t r = r0 + rr;
            if (r >= a.N) break;
            const int c = s_win[rr];
            const float nr = s_nrm[rr];
            const bool counted = !a.row_mask || a.row_mask[r] != 0;
            float ls = 0.f;
            for (int d = lane; d < a.D; d += 64) {
                float g = a.embed[(size_t)c * a.D + d];
                if (a.q_bf16) g = wd_round_bf16(g);
                if (a.q_out) {
                    if (a.q_bf16) ((unsigned short *)a.q_out)[r * a.ldq + d] = wd_f32_to_bf16(g);
                    else ((float *)a.q_out)[r * a.ldq + d] = g;
                }
                if (a.sqerr_partial) {
                    float xv = wd_load<XBF16>(a.x, r * a.ldx + d);
                    if (METRIC == 1 && !a.skip_norm) {
                        xv = xv / nr;
                        if (XBF16) xv = wd_round_bf16(xv);
                    }
                    const float df = g - xv;
                    ls += df * df;
                }
            }
            if (a.sqerr_partial && counted) ds += (double)ls;
        }
    }
    if (a.sqerr_partial) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ds += __shfl_xor(ds, o, 64);
        __syncthreads();
        if (lane == 0) s_red[wave] = ds;
        __syncthreads();
        if (tid == 0) a.sqerr_partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

static int wd_use_mfma()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("VQHIP_WIDE_MFMA"); v = (e && e[0] == '0') ? 0 : 1; }      // =0: the VALU kernel (A/B runs)
    return v;
}

int vq_wide_assign(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed, const float *embed, int C, int metric,
                   int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq, float *best_out, float *rnorm_out, double *sqerr_partial,
                   const uint8_t *row_mask, void *stream)
{
    WideAssignArgs a;
    a.x = x; a.N = N; a.D = D; a.ldx = ldx; a.y2 = packed; a.embed = embed; a.C = C; a.idx_out = idx_out; a.q_out = q_out;
    a.q_bf16 = (q_dtype == VQHIP_BF16); a.ldq = ldq; a.best_out = best_out; a.rnorm_out = rnorm_out; a.sqerr_partial = sqerr_partial;
    a.row_mask = row_mask; a.skip_norm = (metric == VQHIP_COSINE_PRENORM);
    a.vec = (D % 8 == 0) && ((((uintptr_t)x) & 15) == 0) && ((ldx * (x_dtype == VQHIP_BF16 ? 2 : 4)) % 16 == 0) && ((((uintptr_t)embed) & 15) == 0);
    const dim3 grid((unsigned)vqhip_assign_blocks(N));
    hipStream_t st = (hipStream_t)stream;
    const bool bf = x_dtype == VQHIP_BF16;
    if (wd_use_mfma()) {
        static VqAttrOnce o00, o10, o01, o11;
#define WM_LAUNCH(BF, M, ONCE) do { \
            if (int rc = vq_set_max_smem(ONCE, (const void *)vq_wide_mfma_kernel<BF, M>, WM_SMEM, "vq_wide_mfma_kernel")) return rc; \
            hipLaunchKernelGGL((vq_wide_mfma_kernel<BF, M>), grid, dim3(256), WM_SMEM, st, a); } while (0)
        if (metric == VQHIP_EUCLID) {
            if (bf) WM_LAUNCH(true, 0, o10); else WM_LAUNCH(false, 0, o00);
        } else {
            if (bf) WM_LAUNCH(true, 1, o11); else WM_LAUNCH(false, 1, o01);
        }
#undef WM_LAUNCH
        return vq_launch_status("vq_wide_mfma_kernel");
    }
    if (metric == VQHIP_EUCLID) {
        if (bf) hipLaunchKernelGGL((vq_wide_assign_kernel<true, 0>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((vq_wide_assign_kernel<false, 0>), grid, dim3(256), 0, st, a);
    } else {
        if (bf) hipLaunchKernelGGL((vq_wide_assign_kernel<true, 1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((vq_wide_assign_kernel<false, 1>), grid, dim3(256), 0, st, a);
    }
    return vq_launch_status("vq_wide_assign_kernel");
}

// ---- EMA fold of one code row wider than 512 (the arithmetic of ema_embed_row in vqhip.hip: ATen's lerp_, then the division) ---------
__device__ __forceinline__ float wd_lerp(float a, float b, float w)
{
    const float diff = b - a;          // ATen CPU lerp (vectorised form), as aten_lerp in vqhip.hip
    return (fabsf(w) < 0.5f) ? __builtin_fmaf(w, diff, a) : __builtin_fmaf(w - 1.0f, diff, b);
}

__global__ void __launch_bounds__(256) vq_wide_embed_kernel(float *embed_avg, float *embed, const float *embed_sum, const float *weight,
                                                            const float *denom, int C, int D, float omd, int cosine, int do_lerp, int do_update,
                                                            int64_t hs_sum)
{
    const size_t h = blockIdx.y;
    embed_avg += h * (size_t)C * D;
    embed += h * (size_t)C * D;
    if (embed_sum) embed_sum += h * hs_sum;
    if (denom) denom += h * (size_t)C;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const float w = do_lerp ? (weight ? omd * weight[c] : omd) : 0.f;
    float ss = 0.f;
    for (int d = lane; d < D; d += 64) {
        float v = embed_avg[(size_t)c * D + d];
        if (do_lerp) {
            v = wd_lerp(v, embed_sum[(size_t)c * D + d], w);
            embed_avg[(size_t)c * D + d] = v;
        }
        if (do_update) {
            const float e = v / denom[c];
            ss += e * e;
            if (!cosine) embed[(size_t)c * D + d] = e;
        }
    }
    if (do_update && cosine) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float inv = fmaxf(sqrtf(ss), 1e-6f);
        for (int d = lane; d < D; d += 64) embed[(size_t)c * D + d] = (embed_avg[(size_t)c * D + d] / denom[c]) / inv;
    }
}

int vq_wide_ema_embed(float *embed_avg, float *embed, const float *embed_sum, const float *weight, const float *denom, int H, int C, int D,
                      float omd, int cosine, int do_lerp, int do_update, int64_t hs_sum, void *stream)
{
    hipLaunchKernelGGL(vq_wide_embed_kernel, dim3((unsigned)((C + 3) / 4), H), dim3(256), 0, (hipStream_t)stream, embed_avg, embed, embed_sum, weight,
                       denom, C, D, omd, cosine, do_lerp, do_update, hs_sum);
    return vq_launch_status("vq_wide_embed_kernel");
}

// ---- decode: out[n] = sum_q embed_q[idx[n, q]] in q order (rvq.py:525), idx < 0 contributing nothing -----------------------------------
__global__ void __launch_bounds__(256) vq_wide_decode_kernel(const int64_t *__restrict__ idx, int64_t N, int Q, const float *__restrict__ embed,
                                                             int64_t qstride, int C, int D, void *out, int out_bf16, int64_t ldo)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    for (int d = lane; d < D; d += 64) {
        float s = 0.f;
        bool any = false;
        for (int q = 0; q < Q; ++q) {
            const int64_t c = idx[n * Q + q];
            if (c < 0 || c >= C) continue;
            const float v = embed[(size_t)q * qstride + (size_t)c * D + d];
            s = any ? s + v : v;          // (the first stage is copied, not added to +0: keeps a -0)
            any = true;
        }
        if (out_bf16) ((unsigned short *)out)[n * ldo + d] = wd_f32_to_bf16(s);
        else ((float *)out)[n * ldo + d] = s;
    }
}

int vq_wide_decode_sum(const int64_t *idx, int64_t N, int Q, const float *embed, int64_t qstride, int C, int D, void *out, int out_dtype,
                       int64_t ldo, void *stream)
{
    hipLaunchKernelGGL(vq_wide_decode_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, idx, N, Q, embed, qstride, C, D, out,
                       out_dtype == VQHIP_BF16 ? 1 : 0, ldo);
    return vq_launch_status("vq_wide_decode_kernel");
}
