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
static inline size_t wd_align(size_t v, size_t a) { return (v + a - 1) / a * a; }
size_t vq_wide_packed_bytes(int C, int D) { return wd_align((size_t)C * 4, 256) + wd_align((size_t)C * D * 2, 256) + 256; }
static inline size_t wd_bf16_offset(int C) { return wd_align((size_t)C * 4, 256); }

__global__ void __launch_bounds__(256) vq_wide_pack_kernel(const float *embed, int C, int D, float *y2, unsigned short *ebf, size_t in_hs, size_t out_hs)
{
    embed += blockIdx.y * in_hs;
    y2 = (float *)((char *)y2 + blockIdx.y * out_hs);
    ebf = (unsigned short *)((char *)ebf + blockIdx.y * out_hs);
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) y2[c] = wd_aten_sumsq([&](int e) { return embed[(size_t)c * D + e]; }, D);
    const size_t n = (size_t)C * D;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) ebf[e] = wd_f32_to_bf16(embed[e]);
}

int vq_wide_pack(const float *embed, int C, int D, float *packed, int H, void *stream)
{
    if (((uintptr_t)packed) & 15) VQ_FAIL(VQHIP_EALIGN, "pack_codebook: packed must be 16-byte aligned");
    char *base = (char *)packed;
    hipLaunchKernelGGL(vq_wide_pack_kernel, dim3((unsigned)((C + 255) / 256), H), dim3(256), 0, (hipStream_t)stream, embed, C, D, (float *)base,
                       (unsigned short *)(base + wd_bf16_offset(C)), (size_t)C * D, vq_wide_packed_bytes(C, D));
    return vq_launch_status("vq_wide_pack_kernel");
}

// ---- row norms ------------------------------------------------------------------------------------------------------------------
template <bool XBF16>
__global__ void __launch_bounds__(256) vq_wide_sumsq_kernel(const void *x, int64_t N, int D, int64_t ldx, float *out)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    out[n] = wd_aten_sumsq([&](int e) { return wd_load<XBF16>(x, n * ldx + e); }, D);
}

int vq_wide_row_sumsq(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, float *out, void *stream)
{
    const unsigned blocks = (unsigned)((N + 255) / 256);
    if (x_dtype == VQHIP_BF16) hipLaunchKernelGGL(vq_wide_sumsq_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, D, ldx, out);
    else hipLaunchKernelGGL(vq_wide_sumsq_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, D, ldx, out);
    return vq_launch_status("vq_wide_sumsq_kernel");
}

// l2norm of rows (vqp.py:37-38 at :1159): out = x / max(||x||, 1e-6), ||x||^2 in ATen order; bf16 tensors: norm and quotient rounded
// to bf16 like the reference's bf16 ops -- the arithmetic of vqhip_l2norm_rows.  One wave per row; lane 0 sums (sequential order).
template <bool XBF16>
__global__ void __launch_bounds__(256) vq_wide_l2norm_kernel(const void *x, int64_t N, int D, int64_t ldx, void *out, int64_t ldo)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float x2 = 0.f;
    if (lane == 0) x2 = wd_aten_sumsq([&](int e) { return wd_load<XBF16>(x, n * ldx + e); }, D);
    x2 = __shfl(x2, 0, 64);
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

int vq_wide_assign(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed, const float *embed, int C, int metric,
                   int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq, float *best_out, float *rnorm_out, double *sqerr_partial,
                   const uint8_t *row_mask, void *stream)
{
    WideAssignArgs a;
    a.x = x; a.N = N; a.D = D; a.ldx = ldx; a.y2 = packed; a.embed = embed; a.C = C; a.idx_out = idx_out; a.q_out = q_out;
    a.q_bf16 = (q_dtype == VQHIP_BF16); a.ldq = ldq; a.best_out = best_out; a.rnorm_out = rnorm_out; a.sqerr_partial = sqerr_partial;
    a.row_mask = row_mask; a.skip_norm = (metric == VQHIP_COSINE_PRENORM);
    const dim3 grid((unsigned)vqhip_assign_blocks(N));
    hipStream_t st = (hipStream_t)stream;
    const bool bf = x_dtype == VQHIP_BF16;
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
