// vqhip.hip -- hand-written CDNA4 (gfx950, MI355X) kernels behind the C ABI of include/vqhip.h.
//
// Hot path of lucidrains/vector-quantize-pytorch's Codebook.forward (vqp.py:673-791), re-designed
// for wave64 + fp32 MFMA instead of translated from the reference's ATen op sequence:
//
//   vq_pack_kernel      codebook -> MFMA A-operand tiles (+ ||c||^2 in ATen's summation order)
//   vq_assign_kernel    x rows stay resident in VGPRs as the MFMA B operand for the whole codebook
//                       sweep; codebook tiles are streamed L2 -> LDS (register-staged, interleaved with the
//                       MFMAs), double buffered; v_mfma_f32_32x32x2_f32 computes code x row score tiles with
//                       the CODE index on the accumulator-register axis, so the per-row argmin is
//                       lane-local (no cross-lane reduce inside the sweep) and the N x C distance
//                       matrix is never written; gather + squared-error partials in the epilogue.
//   vq_rvq_kernel       the same sweep repeated over Q residual stages with the residual updated in the
//                       resident B-operand registers (ResidualVQ.forward's loop in one launch)
//   vq_hist / vq_scan / vq_scatter / vq_segsum_kernel
//                       EMA sufficient statistics: counting sort of the rows by code, then one workgroup
//                       per (code, row-segment) sums full rows -- no fp32 atomics on the hot part
//   vq_ema_*_kernel     lerp fold, Laplace smoothing, codebook renormalisation
//   vq_route_kernel     straight-through / rotation-trick output and its backward (+ commit-loss grad)
//   vq_decode_kernel    indices -> (summed) codes
//   vq_reduce_kernel    deterministic fp64 reduction of the per-block loss partials
//
// Numerics contract (see DESIGN.md "bit-exact indices"): every fp32 rounding after the dot product
// follows the reference's CPU arithmetic; the dot product itself is ONE fp32 FMA chain in ascending
// k (what the f32 MFMA computes), which oracle/vq_oracle.c restates.  BUILD WITH -ffp-contract=off.

#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <type_traits>

#include "vqhip_internal.h"
#include "vq_route_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

thread_local char vq_g_err[256] = "";
#define g_err vq_g_err

extern "C" const char *vqhip_last_error(void) { return g_err; }
extern "C" const char *vqhip_version(void) { return "vqhip 0.2 (gfx950)"; }

int vq_launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
#define launch_status vq_launch_status

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ float round_to_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_rne(f)); }

template <bool BF16>
__device__ __forceinline__ float load_elem(const void *base, int64_t off)
{
    if (BF16) return bf16_bits_to_f32(((const unsigned short *)base)[off]);
    return ((const float *)base)[off];
}

// ATen CPU order of sum(x*x) over one contiguous row of D <= 512 floats (vqp.py:59-60).
// 32 interleaved chains (8 SIMD lanes x 4 ILP accumulators), leftovers and tail as in
// oracle/vq_oracle.c::aten_sumsq_row.  Sequential, one thread per row: only used for the codebook
// and for odd D; the assign kernel has an in-register version for D % 32 == 0.
template <typename F>
__device__ float aten_sumsq_seq(F ld, int D)
{
    const int V = D >> 3;
    const int size = V >> 2;
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    for (int i = 0; i < size; ++i) {
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            float v = ld(32 * i + c);
            acc[c] += v * v;
        }
    }
    for (int v = size * 4; v < V; ++v) {
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float t = ld(v * 8 + l);
            acc[l] += t * t;
        }
    }
    float fin = 0.f;
    for (int e = V * 8; e < D; ++e) {
        float t = ld(e);
        fin += t * t;
    }
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        float p = ((acc[l] + acc[8 + l]) + acc[16 + l]) + acc[24 + l];
        fin += p;
    }
    return fin;
}

// ------------------------------------------------------------------------------------------------
// codebook packing.  Tile t (32 codes) = 32*DT floats in A-operand order + 256 floats of which the
// first 32 are ||c||^2 (+inf for padding codes):
//   tile[((t4*2 + hi)*32 + i)*4 + jj] = embed[t*32 + i][8*t4 + 2*jj + hi]
// so that lane (i, hi) fetches the A values of MFMA k-steps 4*t4 .. 4*t4+3 with one ds_read_b128
// and a whole wave reads 1 KiB contiguous (conflict-free).
// ------------------------------------------------------------------------------------------------
#define pick_dt vq_pick_dt
#define packed_bf16_offset vq_packed_bf16_offset

// The same sum with the 32 chains on the 32 lanes of a half-wave (lane c = chain c; `ld` is called with the same element by no two lanes
// of the chain phase, the loads of a step are one contiguous 128-byte piece of the row); the result is returned in every lane of the
// half-wave.  All 32 lanes must call.  (vq_wide.hip holds the same routine for D > 512, with the cascade level those sizes add.)
template <typename F>
__device__ __forceinline__ float aten_sumsq_coop(F ld, int D, int c)
{
    const int V = D >> 3;
    const int size = V >> 2;
    float a0 = 0.f;
    for (int i = 0; i < size; ++i) {
        const float v = ld(32 * i + c);
        a0 += v * v;
    }
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
    const int base = (int)(threadIdx.x & 32);
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float p = ((__shfl(a0, base + l, 64) + __shfl(a0, base + 8 + l, 64)) + __shfl(a0, base + 16 + l, 64)) + __shfl(a0, base + 24 + l, 64);
        fin += p;
    }
    return fin;
}

// regions of 32-bit words the fused train step wants zeroed before its search (statistics, histograms, list headers): zeroed by spare
// workgroups of the pack kernel (blockIdx.y == 2) instead of a launch of their own
#define VQ_STEP_MAX_CHUNKS 4
#define VQ_ZERO_REGIONS (1 + 2 * VQ_STEP_MAX_CHUNKS)
struct ZeroArgs { unsigned *p[VQ_ZERO_REGIONS]; unsigned n[VQ_ZERO_REGIONS]; };   // regions of n 32-bit words each (n = 0: unused)

extern "C" size_t vqhip_packed_bytes(int C, int D)
{
    if (C > 0 && vq_is_wide(D)) return vq_wide_packed_bytes(C, D);     // y2 || bf16 copy (vq_wide.hip)
    const int DT = pick_dt(D);
    if (DT == 0 || C <= 0) return 0;
    return vq_packed_total_bytes(C, D);   // layout: vqhip_internal.h
}

__global__ void __launch_bounds__(256) vq_pack_kernel(const float *embed, int C, int D, int DT,
                                                      float *packed, unsigned short *ebf,
                                                      unsigned *scalars, size_t head_bytes, const ZeroArgs z)
{
    __shared__ float y2sh[32];
    const int t = blockIdx.x;
    if (blockIdx.y == 2) {      // the fused train step's zeroing (one set of regions: head 0's workgroups only)
        if (blockIdx.z) return;
#pragma unroll
        for (int r = 0; r < VQ_ZERO_REGIONS; ++r)
            for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < z.n[r]; i += gridDim.x * 256) z.p[r][i] = 0u;
        return;
    }
    if (blockIdx.z) {      // batched heads (vqhip_pack_codebook_batched): head z's codebook and packed buffer
        embed += (size_t)blockIdx.z * C * D;
        packed = (float *)((char *)packed + (size_t)blockIdx.z * head_bytes);
        ebf = (unsigned short *)((char *)ebf + (size_t)blockIdx.z * head_bytes);
        scalars = (unsigned *)((char *)scalars + (size_t)blockIdx.z * head_bytes);
    }
    // blockIdx.y splits a tile's work over two workgroups (a 1024-code codebook is only 32 tiles): y = 1 writes the bf16 copy,
    // y = 0 the fp32 tile and the norms
    if (blockIdx.y == 1) {
    if ((D & 3) == 0) {   // RNE-rounded bf16 copy of this tile's 32 code rows, 4 elements per store
        for (int p = threadIdx.x; p < 8 * D; p += 256) {
            const size_t o = (size_t)t * 32 * D + 4 * (size_t)p;
            if (o < (size_t)C * D) {
                const f32x4 v = *(const f32x4 *)(embed + o);
                uint2 w;
                w.x = (unsigned)f32_to_bf16_rne(v.x) | ((unsigned)f32_to_bf16_rne(v.y) << 16);
                w.y = (unsigned)f32_to_bf16_rne(v.z) | ((unsigned)f32_to_bf16_rne(v.w) << 16);
                *(uint2 *)(ebf + o) = w;
            }
        }
    } else {
        for (int p = threadIdx.x; p < 32 * D; p += 256) {
            const size_t o = (size_t)t * 32 * D + p;
            if (o < (size_t)C * D) ebf[o] = f32_to_bf16_rne(embed[o]);
        }
    }
    return;
    }
    const int tile_f = 32 * DT + 256;
    float *out = packed + (size_t)t * tile_f;
    // The tile's code rows go through LDS: read from the codebook the way they lie (whole rows, 16 bytes per lane), written in the
    // A-operand order 256 contiguous bytes per 16 codes, and the norms summed from the LDS copy.  (Reading the codebook in tile order --
    // 4-byte loads 1 KiB apart -- and one thread per norm made this kernel 12 us of the 0.83 ms cfg-2 step.)  G codes at a time:
    // 32 rows of <= 256 floats or 16 of <= 512, row stride D + 1 floats (the tile-order reads walk down the rows).
    __shared__ float sh[32 * 257];
    const int G = D <= 256 ? 32 : 16, LD = D + 1;
    const bool vec = (D & 3) == 0 && (((uintptr_t)embed) & 15) == 0;
#pragma unroll 1
    for (int g0 = 0; g0 < 32; g0 += G) {
        if (g0) __syncthreads();
        if (vec) {
            const int D4 = D >> 2;
            for (int q = threadIdx.x; q < G * D4; q += 256) {
                const int il = q / D4, k = (q - il * D4) * 4, code = t * 32 + g0 + il;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (code < C) v = *(const f32x4 *)(embed + (size_t)code * D + k);
                float *d = sh + il * LD + k;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int q = threadIdx.x; q < G * D; q += 256) {
                const int il = q / D, k = q - il * D, code = t * 32 + g0 + il;
                sh[il * LD + k] = code < C ? embed[(size_t)code * D + k] : 0.f;
            }
        }
        __syncthreads();
        for (int q = threadIdx.x; q < G * DT; q += 256) {
            const int jj = q & 3, il = (q >> 2) % G, rest = q / (4 * G), hi = rest & 1, t4 = rest >> 1;
            const int k = 8 * t4 + 2 * jj + hi;
            out[((t4 * 2 + hi) * 32 + g0 + il) * 4 + jj] = k < D ? sh[il * LD + k] : 0.f;      // (rows past C were staged as zeros)
        }
        // ||c||^2 in ATen's order, a half-wave per code (eight codes at a time)
        for (int r8 = 0; r8 < G; r8 += 8) {
            const int il = r8 + (threadIdx.x >> 5), code = t * 32 + g0 + il;
            const float *r = sh + il * LD;
            float v = aten_sumsq_coop([&](int e) { return r[e]; }, D, (int)(threadIdx.x & 31));
            if (code >= C) v = INFINITY;
            if ((threadIdx.x & 31) == 0) y2sh[g0 + il] = v;
        }
    }
    __syncthreads();
    {
        // tail: floats [0, 32) ||c||^2, float [32] the largest of the tile's real codes as float bits (non-negative floats order like their
        // bit patterns; NaN sorts above inf) -- vq_pack16_kernel takes the maximum over the tiles: no atomic, nothing to zero beforehand
        const int i = threadIdx.x;
        float v = i < 32 ? y2sh[i] : 0.f;
        if (i == 32) {
            unsigned m = 0u;
            for (int k = 0; k < 32; ++k)
                if (t * 32 + k < C) m = max(m, __float_as_uint(y2sh[k]));
            v = __uint_as_float(m);
        }
        out[32 * DT + i] = v;
    }
    if (t == 0 && threadIdx.x == 0) { scalars[1] = 0u; scalars[4] = 0u; }    // the maxima vq_pack16_kernel accumulates with atomics
}

// Second pack phase (needs max ||c||^2 of the first): fp16 A-operand tiles of the single-pass screening kernel
// (vq_screen.hip, vq_screen16_kernel).  The codebook is scaled by 2^sc so that max ||c|| lands in [2^13, 2^14) -- every
// element then is below 2^14 and fp16 keeps 11 significant bits down to 2^-28 of the largest possible element -- and
// rounded to fp16 (RNE); the kernel compensates the scale exactly (powers of two).  Same lane order as the bf16 tiles:
// 16 bytes per lane and k-step; lane l = code (l & 31), k-slot 8 * (l >> 5) + e.  Tile tail: 32 floats -||c||^2 / 2
// (-3e38 for padding codes), then 32 floats ||c|| (round 6).  scalars[1] <- rho, scalars[3] <- r0 with ||c - c_f16|| <= rho ||c|| + r0
// for every code (the certificate charges X * that for the rounding), scalars[2] <- sc, scalars[4] <- ~bits(min ||c||^2).
__global__ void __launch_bounds__(256) vq_pack16_kernel(const float *embed, int C, int D, int DT, int n_tiles,
                                                        const float *packed, char *tiles16,
                                                        unsigned *scalars, size_t head_bytes)
{
    __shared__ float rsq[32];
    const int t = blockIdx.x;
    if (blockIdx.z) {
        embed += (size_t)blockIdx.z * C * D;
        packed = (const float *)((const char *)packed + (size_t)blockIdx.z * head_bytes);
        tiles16 += (size_t)blockIdx.z * head_bytes;
        scalars = (unsigned *)((char *)scalars + (size_t)blockIdx.z * head_bytes);
    }
    // max ||c||^2 over the codebook: the largest of the tile maxima vq_pack_kernel left behind the fp32 tiles
    __shared__ unsigned ymax_s;
    if (threadIdx.x == 0) ymax_s = 0u;
    __syncthreads();
    {
        unsigned m = 0u;
        for (int tt = threadIdx.x; tt < n_tiles; tt += 256) m = max(m, __float_as_uint(packed[(size_t)tt * (32 * DT + 256) + 32 * DT + 32]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        if ((threadIdx.x & 63) == 0 && m) atomicMax(&ymax_s, m);
    }
    __syncthreads();
    const unsigned y2bits = ymax_s;
    int sc = 0;
    if (y2bits != 0u) {
        const int e2 = (int)((y2bits >> 23) & 0xffu) - 127;   // max ||c||^2 in [2^e2, 2^(e2+1))  =>  max ||c|| < 2^((e2 >> 1) + 1)
        sc = 13 - (e2 >> 1);
        sc = sc < -100 ? -100 : (sc > 100 ? 100 : sc);
    }
    const float Sc = __uint_as_float((unsigned)(sc + 127) << 23), iSc = __uint_as_float((unsigned)(127 - sc) << 23);
    if (threadIdx.x < 32) rsq[threadIdx.x] = 0.f;
    __syncthreads();
    _Float16 *st = (_Float16 *)(tiles16 + (size_t)t * vq_tile16_bytes(DT));
    for (int p = threadIdx.x; p < 4 * DT; p += 256) {   // one (k-step, lane) per iteration: 8 features -> one fp16 fragment
        const int l = p & 63;
        const int ks = p >> 6;
        const int code = t * 32 + (l & 31);
        const int k0 = ks * 16 + 8 * (l >> 5);
        float v[8];
        if (code < C && k0 + 8 <= D && (D & 3) == 0) {
            const f32x4 a0 = *(const f32x4 *)(embed + (size_t)code * D + k0), a1 = *(const f32x4 *)(embed + (size_t)code * D + k0 + 4);
            v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (code < C && k0 + e < D) ? embed[(size_t)code * D + k0 + e] : 0.f;
        }
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        f16x8 h;
        float r2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 he = (_Float16)(v[e] * Sc);          // the scaling is exact, the conversion rounds to nearest even
            h[e] = he;
            const float d = v[e] - (float)he * iSc;              // exact difference of two fp32 values that agree to 11 bits
            r2 = __builtin_fmaf(d, d, r2);
        }
        *(f16x8 *)(st + ((size_t)ks * 64 + l) * 8) = h;
        if (code < C) atomicAdd(&rsq[l & 31], r2);
    }
    __syncthreads();
    {
        // tile tail (1 KiB): floats [0, 32) the accumulator's start value, [32, 64) an upper bound of ||c|| per code -- the screening
        // kernels add (row factor) x that to the start value, so that every score they track is an UPPER bound of the code's true
        // score and the certificate only ever charges a row for the codes it is actually compared with (round 6: the bound used to be
        // built from the codebook-wide max ||c||^2 and max ||c - c_h||, so ONE large-norm code raised the threshold of every row).
        //   start value  -||c||^2 / 2 + kb ||c||^2,  kb = u (5 + 1.001 (DT + 1) + 0.51) 1.001: the code's own share of the terms that
        //   do not depend on the row (sqrt collapse 5 u y2, MFMA accumulation of the start value, the rounding of the kernel's FMA)
        float *nh = (float *)((char *)st + (size_t)64 * DT);
        const int i = threadIdx.x;
        const bool real = (i < 32) && (t * 32 + i < C);
        // ||c||^2 of the tile's codes sits behind the fp32 tile of the exact section (written by vq_pack_kernel)
        const float y2 = (real && t < n_tiles) ? packed[(size_t)t * (32 * DT + 256) + 32 * DT + i] : 0.f;
        const float kb = 5.9604645e-8f * (5.f + 1.001f * (float)(DT + 1) + 0.51f) * 1.001f;
        const float yb = sqrtf(y2) * 1.0001f;                   // (y2 is ATen's fp32 sum: off the true norm by < D u relative)
        // (a norm that overflowed fp32 -- y2 = inf / NaN / >= 1e37 -- gets finite placeholders: no NaN may enter the top-3 fold, which
        //  would silently drop it; the screening kernels certify NOTHING against such a codebook, scalars[0])
        const bool sane = y2 < 1e37f;
        if (i < 32) {
            nh[i] = real ? (sane ? __builtin_fmaf(kb, y2, -0.5f * y2) : -3.0e38f) : -3.0e38f;
            nh[32 + i] = (real && sane) ? fminf(yb, 1e19f) : 0.f;
            nh[64 + i] = (real && sane) ? -0.5f * y2 : -3.0e38f;      // the plain start value (debug view of the screening kernels only)
        } else if (i < 192) nh[64 + i] = 0.f;
        // the rounding of the fp16 copy, per code: every element errs by <= max(2^-12 |c_i|, 2^(-25 - sc)) (RNE; subnormal spacing below
        // 2^-14 scaled), so ||c - c_h|| <= rho ||c|| + r0 with r0 = sqrt(DT) 2^(-25 - sc) and rho <= 2^-12: rho is MEASURED as the
        // largest (||c - c_h|| - r0)+ / ||c|| -- a relative quantity that no single code's size can inflate
        const float r0 = sqrtf((float)DT) * __uint_as_float((unsigned)(127 - 25 - sc) << 23) * 1.01f;
        if (real) {
            const float r = sqrtf(rsq[i]) * 1.01f + 1e-38f;     // slack for the fp32 rounding of the sum above
            const float over = r - r0;
            if (over > 0.f && y2 > 0.f) atomicMax(scalars + 1, __float_as_uint(fminf(over / sqrtf(y2) * 1.001f, 2.5e-4f)));
            atomicMax(scalars + 4, ~__float_as_uint(y2));       // ~(bits of the SMALLEST ||c||^2): the scalars start zeroed; non-negative floats
        }                                                       // order like their bits (NaN sorts above inf: read back as "no plain mode")
        if (t == 0 && i == 0) { scalars[0] = y2bits; scalars[2] = (unsigned)sc; scalars[3] = __float_as_uint(r0); }
    }
}

static int pack_codebook_impl(const float *embed, int C, int D, float *packed, void *stream, int H = 1, const ZeroArgs *zero = nullptr)
{
    if (!embed || !packed || C <= 0 || H < 1) VQ_FAIL(VQHIP_EINVAL, "pack_codebook: null pointer, C <= 0 or H < 1");
    if (vq_is_wide(D)) return vq_wide_pack(embed, C, D, packed, H, stream);
    const int DT = pick_dt(D);
    if (D < 1 || DT == 0) VQ_FAIL(VQHIP_EDIM, "pack_codebook: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (((uintptr_t)packed) & 15) VQ_FAIL(VQHIP_EALIGN, "pack_codebook: packed must be 16-byte aligned");
    const int tiles = (C + 31) / 32;
    char *base = (char *)packed;
    unsigned *scalars = (unsigned *)(base + vq_packed_scalars_offset(C, D));
    const size_t head_bytes = vq_packed_total_bytes(C, D);     // (a multiple of 16: every head's buffer keeps the alignment)
    ZeroArgs z;
    if (zero) z = *zero;
    else for (int r = 0; r < VQ_ZERO_REGIONS; ++r) { z.p[r] = nullptr; z.n[r] = 0; }
    hipLaunchKernelGGL(vq_pack_kernel, dim3(tiles, zero ? 3 : 2, H), dim3(256), 0, (hipStream_t)stream, embed, C, D, DT, packed,
                       (unsigned short *)(base + packed_bf16_offset(C, D)), scalars, head_bytes, z);
    if (int rc = launch_status("vq_pack_kernel")) return rc;
    hipLaunchKernelGGL(vq_pack16_kernel, dim3((unsigned)vq_tiles16(C), 1, H), dim3(256), 0, (hipStream_t)stream, embed, C, D, DT, tiles,
                       (const float *)packed, base + vq_packed_f16_offset(C, D), scalars, head_bytes);
    return launch_status("vq_pack16_kernel");
}

extern "C" int vqhip_pack_codebook(const float *embed, int C, int D, float *packed, void *stream)
{
    return pack_codebook_impl(embed, C, D, packed, stream);
}

extern "C" int vqhip_pack_codebook_batched(const float *embed, int H, int C, int D, float *packed, void *stream)
{
    return pack_codebook_impl(embed, C, D, packed, stream, H);
}

// ------------------------------------------------------------------------------------------------
// assignment
// ------------------------------------------------------------------------------------------------
// The DT/2 MFMAs of one 32-code tile, with the copy of the NEXT tile (L2 -> LDS) interleaved.
// The copy is register-staged in 3 batches: `gsrc` / `ldst` point at this lane's 16 bytes of piece 0 of the next
// tile (global / LDS); piece k of this wave is 4 KiB further (waves interleave 1-KiB pieces).  Loads of a batch
// are issued behind one MFMA, the ds_write_b128s a few MFMA groups later when the data has arrived.
// Why not LDS-DMA (global_load_lds) as in the first version: each glds kept the issuing wave busy ~270 cycles
// (9 per tile = 2.4k cycles per wave and tile, measured in round 1), and because a wave's MFMAs form ONE dependent
// chain nothing of that hides behind its own MFMAs; a global_load + ds_write pair costs the wave ~30 cycles.
template <int DT>
__device__ __forceinline__ void mfma_sweep_tile(const f32x4 *ap, const float (&xr)[DT / 2], f32x16 &acc,
                                                const char *gsrc, char *ldst, int npieces)
{
    constexpr int NG = DT / 8;
    constexpr int NCHUNK = (32 * DT + 256) * 4 / 1024;
    constexpr int PMAX = (NCHUNK + 3) / 4;              // pieces per wave (wave 0 may own one more than wave 3)
    constexpr int NB = (NG >= 12) ? 3 : ((NG >= 4) ? 2 : 1);   // batches
    constexpr int BS = (PMAX + NB - 1) / NB;            // pieces per batch
    constexpr int SP = (NG * 3 / 4) / NB > 0 ? (NG * 3 / 4) / NB : 1;   // t4 steps between batch starts
    constexpr int LAG = (SP * 2 / 3) > 0 ? (SP * 2 / 3) : 1;            // t4 steps between a batch's loads and its stores
    f32x4 stg[BS];
#pragma unroll
    for (int t4 = 0; t4 < NG; ++t4) {
        const f32x4 av = ap[t4 * 64];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, xr[4 * t4 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, xr[4 * t4 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, xr[4 * t4 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, xr[4 * t4 + 3], acc, 0, 0, 0);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (t4 == b * SP) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < BS; ++i)   // UNCONDITIONAL (a guarded load makes hipcc wait vmcnt(0) right behind it);
                    if (b * BS + i < PMAX)     // a piece beyond this wave's share reads into the 4 KiB tail pad at worst
                        stg[i] = *(const f32x4 *)(gsrc + (size_t)(b * BS + i) * 4096);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t4 == b * SP + LAG) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < BS; ++i)
                    if (b * BS + i < npieces) *(f32x4 *)(ldst + (b * BS + i) * 4096) = stg[i];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// Per-tile argmin epilogue.  acc[4q + r] = <code base + 8q + 4hi + r, row j>: 16 scores of ONE row per lane,
// codes ascending with the register index, tiles ascending, so "first minimum" == lowest code index.
//
// Cost model (measured with s_memtime stamps, tools/trace_tiles.py): v_mfma_f32_32x32x2_f32 runs on the SIMD's fp32
// lanes, so every VALU instruction of either resident wave takes ~7 cycles AWAY from the MFMA stream -- VALU work is
// not hidden behind fp32 MFMAs, it is serialised with them.  The epilogue is therefore written for minimum VALU
// instruction COUNT (about 90 per tile; the first version, one wave-level branch per element, had ~190 + 16 branches):
//   1. s = (x2 + y2) + (-2 xy)          v_pk_add_f32 + v_pk_fma_f32, two codes per instruction  (vqp.py:62, one
//                                        rounding per operation exactly like the reference; the product by -2 is exact)
//   2. s_min                             v_min3_f32 tree; clamp(min=1e-8) applied to the minimum only (monotone)
//   3. skip the tile (wave-uniform) unless some lane has s_min < s_best: sqrt is monotone, it cannot win
//   4. d = sqrt_rn(s_min)                ONE correctly rounded sqrt per lane and tile
//   5. the reference compares d, and the sqrt collapses neighbouring s: the winner is the FIRST element whose sqrt
//      rounds to d.  {s : sqrt_rn(s) <= d} = {s < m^2}, m = midpoint of d and the next float; m has 25 significant
//      bits so m^2 is exact in double and never equals an fp32 value: U = largest fp32 below m^2, then a descending
//      scan "s_e <= U ? e" leaves the first such element.  Exact, no per-element sqrt, no per-element branch.
template <int METRIC>
__device__ __forceinline__ void argmin_tile(const f32x16 &acc, const float *y2s, float x2, int base, int hi, int C,
                                            float &bd, float &bs, int &bi)
{
    if (METRIC != 0) {   // cosine: first maximum of the raw dot products (vqp.py:741, 140)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int code = base + 8 * (e >> 2) + 4 * hi + (e & 3);
            const bool win = (acc[e] > bd) && (code < C);
            bd = win ? acc[e] : bd;
            bi = win ? code : bi;
        }
        return;
    }
    f32x2 s2[8];
    const f32x2 xx = {x2, x2};
    const f32x2 m2c = {-2.0f, -2.0f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 yv = *(const f32x4 *)(y2s + 8 * q + 4 * hi);
        const f32x2 y01 = {yv.x, yv.y}, y23 = {yv.z, yv.w};
        const f32x2 a01 = {acc[4 * q + 0], acc[4 * q + 1]}, a23 = {acc[4 * q + 2], acc[4 * q + 3]};
        s2[2 * q + 0] = __builtin_elementwise_fma(m2c, a01, xx + y01);
        s2[2 * q + 1] = __builtin_elementwise_fma(m2c, a23, xx + y23);
    }
    float s[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = s2[e >> 1][e & 1];
    // v_min3_f32 tree
    float m0 = fminf(fminf(s[0], s[1]), s[2]), m1 = fminf(fminf(s[3], s[4]), s[5]), m2 = fminf(fminf(s[6], s[7]), s[8]);
    float m3 = fminf(fminf(s[9], s[10]), s[11]), m4 = fminf(fminf(s[12], s[13]), s[14]);
    m0 = fminf(fminf(m0, m1), m2);
    m3 = fminf(fminf(m3, m4), s[15]);
    const float sm = fmaxf(fminf(m0, m3), 1e-8f);   // clamp(min = 1e-8) commutes with the minimum
    const bool improving = sm < bs;                  // sm >= bs  =>  sqrt(sm) >= bd: this tile cannot win for this lane
    if (!__any(improving)) return;
    const float d = sqrtf(sm);
    const bool win = improving && (d < bd);
    if (!__any(win)) return;
    const double mid = 0.5 * ((double)d + (double)__uint_as_float(__float_as_uint(d) + 1u));
    const double mid2 = mid * mid;
    float U = (float)mid2;                           // round to nearest ...
    U = ((double)U > mid2) ? __uint_as_float(__float_as_uint(U) - 1u) : U;   // ... then down: largest fp32 < m^2
    int ew = 15;
#pragma unroll
    for (int e = 14; e >= 0; --e) ew = (s[e] <= U) ? e : ew;   // s[15] <= U is implied when nothing earlier matches
    bd = win ? d : bd;
    bs = win ? sm : bs;
    bi = win ? (base + 8 * (ew >> 2) + 4 * hi + (ew & 3)) : bi;
}

// ||x||^2 of the wave's rows in ATen's CPU order (vqp.py:59), from the "load layout" registers
// xr[4m + r] = x[row][8m + 4hi + r]: chain (e % 32) over e ascending, then lane-wise ((a0+a1)+a2)+a3 over the ILP
// accumulators, then the 8 SIMD lanes left to right.  Element e = 8m + 4hi + r sits in chain [m & 3][4hi + r].
// Zero padding (D < DT) only adds +0 to non-negative chains.  Valid for D % 32 == 0 (odd D: pre-pass, see host).
template <int DT>
__device__ __forceinline__ float x2_aten_order(const float (&xr)[DT / 2], int j)
{
    constexpr int NG = DT / 8;
    // two chains per v_pk_mul_f32 / v_pk_add_f32 (each component rounds exactly like the scalar op)
    f32x2 ch[4][2];
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) { ch[mm][0] = f32x2{0.f, 0.f}; ch[mm][1] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int m = 0; m < NG; ++m) {
        const f32x2 v01 = {xr[4 * m + 0], xr[4 * m + 1]}, v23 = {xr[4 * m + 2], xr[4 * m + 3]};
        ch[m & 3][0] = ch[m & 3][0] + v01 * v01;
        ch[m & 3][1] = ch[m & 3][1] + v23 * v23;
    }
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = ((ch[0][r >> 1][r & 1] + ch[1][r >> 1][r & 1]) + ch[2][r >> 1][r & 1]) + ch[3][r >> 1][r & 1];
    const float f_lo = ((p[0] + p[1]) + p[2]) + p[3];          // SIMD lanes 0..3 (hi == 0 half)
    const float f_from_lo = __shfl(f_lo, j, 64);               // value of the hi == 0 partner
    const float f_hi = (((f_from_lo + p[0]) + p[1]) + p[2]) + p[3];  // continue with lanes 4..7
    return __shfl(f_hi, j + 32, 64);
}

#ifdef VQ_TRACE
long long *vq_g_trace = nullptr;
#define g_trace vq_g_trace
extern "C" void vqhip_set_trace(long long *p) { g_trace = p; }
#define VQ_STAMP(slot) do { if (a.trace && blockIdx.x < 16 && lane == 0 && ct < 64) a.trace[(((size_t)blockIdx.x * 4 + wave) * 64 + ct) * 4 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define VQ_STAMP(slot) do {} while (0)
#endif

struct AssignArgs {
    const void *x;
    int64_t N;
    int D;
    int64_t ldx;
    const float *packed;
    const float *embed;
    const unsigned short *embed_bf16;  // RNE-rounded copy inside the packed buffer
    int C;
    int n_tiles;
    int64_t *idx_out;
    void *q_out;
    int q_bf16;
    int64_t ldq;
    float *best_out;
    float *rnorm_out;
    double *sqerr_partial;
    const uint8_t *row_mask;
    float *scores_out;  // nullable [N, lds]: the reference's `dist` tensor (-cdist or cosine similarity), rare options only
    int64_t lds;
    // streaming log-sum-exp over the score row (cross-entropy to codes, vqp.py:1242-1256) instead of materialising it:
    float *lse_out;            // nullable [N]: log sum_c exp(dist[n, c])
    float *tscore_out;         // nullable [N]: dist[n, target[n]] (target null: the winner's score)
    const int64_t *target;     // nullable [N]; negative = ignored row (tscore 0)
    int x_vec;  // 1: D == DT and x rows are vector-load aligned
    int q_vec;  // 1: D == DT and q rows are vector-store aligned
    int skip_norm;  // cosine: rows are already unit-norm
    // several heads in one launch (vqhip_assign_batched: blockIdx.y = head): byte strides between consecutive heads' buffers
    // (index / q / rnorm outputs only then)
    int heads;
    int64_t hs_x, hs_packed, hs_embed, hs_idx, hs_q, hs_rnorm;
#ifdef VQ_TRACE
    long long *trace;
#endif
};

__device__ __forceinline__ void swap32(float &a, float &b)
{
    // v_permlane32_swap: lanes 32..63 of `a` <-> lanes 0..31 of `b`
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

template <int DT, bool XBF16, int METRIC>
__global__ void __launch_bounds__(256, (DT <= 256 ? 2 : 1)) vq_assign_kernel(const AssignArgs a0)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AssignArgs a = a0;
    if (a0.heads > 1) {
        const int64_t h = blockIdx.y;
        a.x = (const char *)a0.x + h * a0.hs_x;
        a.packed = (const float *)((const char *)a0.packed + h * a0.hs_packed);
        a.embed_bf16 = (const unsigned short *)((const char *)a0.embed_bf16 + h * a0.hs_packed);
        a.embed = (const float *)((const char *)a0.embed + h * a0.hs_embed);
        a.idx_out = (int64_t *)((char *)a0.idx_out + h * a0.hs_idx);
        if (a0.q_out) a.q_out = (char *)a0.q_out + h * a0.hs_q;
        if (a0.rnorm_out) a.rnorm_out = (float *)((char *)a0.rnorm_out + h * a0.hs_rnorm);
    }
    constexpr int TILE_F = 32 * DT + 256;
    constexpr int TILE_B = TILE_F * 4;
    constexpr int NCHUNK = TILE_B / 1024;  // 1 KiB LDS-DMA pieces per tile
    constexpr int NG = DT / 8;             // groups of 8 consecutive features

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;   // row within the wave's 32-row block  (MFMA N index)
    const int hi = lane >> 5;  // which of the 2 k's of an MFMA step this lane feeds
    const int64_t row = (int64_t)blockIdx.x * VQHIP_ASSIGN_ROWS_PER_BLOCK + wave * 32 + j;
    const bool row_ok = row < a.N;
    const int64_t rowc = row_ok ? row : (a.N - 1);

    // ---- codebook tiles L2 -> LDS: wave w moves the 1-KiB pieces w, w+4, ... of a tile (this lane: 16 bytes of each) ----
    const int my_pieces = (NCHUNK - wave + 3) / 4;
    const int piece_off = wave * 1024 + lane * 16;
    for (int k = 0; k < my_pieces; ++k)   // tile 0, while x is being loaded
        *(f32x4 *)(smem + piece_off + k * 4096) = *(const f32x4 *)((const char *)a.packed + piece_off + (size_t)k * 4096);

    // ---- x rows -> registers.  xr[4m + r] = x[row][8m + 4hi + r]  ("load layout") ----------------
    float xr[DT / 2];
    if (a.x_vec) {
        if (XBF16) {
            const uint2 *p = (const uint2 *)((const unsigned short *)a.x + rowc * a.ldx + 4 * hi);
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                const uint2 w = p[m * 2];
                xr[4 * m + 0] = __uint_as_float(w.x << 16);
                xr[4 * m + 1] = __uint_as_float(w.x & 0xffff0000u);
                xr[4 * m + 2] = __uint_as_float(w.y << 16);
                xr[4 * m + 3] = __uint_as_float(w.y & 0xffff0000u);
            }
        } else {
            const f32x4 *p = (const f32x4 *)((const float *)a.x + rowc * a.ldx + 4 * hi);
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                const f32x4 w = p[m * 2];
                xr[4 * m + 0] = w.x;
                xr[4 * m + 1] = w.y;
                xr[4 * m + 2] = w.z;
                xr[4 * m + 3] = w.w;
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < NG; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 8 * m + 4 * hi + r;
                xr[4 * m + r] = (k < a.D) ? load_elem<XBF16>(a.x, rowc * a.ldx + k) : 0.f;
            }
    }

    // ---- ||x||^2 in ATen's order (vqp.py:59): chain (e % 32) over e ascending, then lane-wise
    //      ((a0+a1)+a2)+a3 over the ILP accumulators, then the 8 SIMD lanes left to right.
    //      In the load layout element e = 8m + 4hi + r sits in chain [m & 3][4hi + r].
    //      Zero padding (D < DT) only adds +0 to non-negative chains.  Rows whose D is not a
    //      multiple of 32 take x2 from a pre-pass (a.rnorm_out pre-filled) -- see host wrapper. ----
    float x2 = x2_aten_order<DT>(xr, j);
    if (a.D & 31) x2 = a.rnorm_out[rowc];  // exact ATen order for odd D was precomputed

    float nrm = 0.f;
    if (METRIC == 1 && a.skip_norm) nrm = 1.f;
    if (METRIC == 1 && !a.skip_norm) {
        // l2norm (vqp.py:37-38): x / max(||x||, 1e-6); for bf16 inputs the reference normalises in
        // bf16 (norm and quotient both rounded to bf16) before Codebook.forward casts to fp32.
        nrm = sqrtf(x2);
        if (XBF16) nrm = round_to_bf16(nrm);
        nrm = fmaxf(nrm, XBF16 ? round_to_bf16(1e-6f) : 1e-6f);
#pragma unroll
        for (int q = 0; q < DT / 2; ++q) {
            float v = xr[q] / nrm;
            xr[q] = XBF16 ? round_to_bf16(v) : v;
        }
    }

    // ---- load layout -> MFMA B-operand layout: after the swaps register 4m+{0,2,1,3} holds
    //      x[row][8m + 2*jj + hi] for jj = 0..3, i.e. the k = 2t + hi element of k-step t = 4m + jj.
#pragma unroll
    for (int m = 0; m < NG; ++m) {
        swap32(xr[4 * m + 0], xr[4 * m + 1]);
        swap32(xr[4 * m + 2], xr[4 * m + 3]);
    }

    // ---- sweep the codebook ----------------------------------------------------------------------
    float bd = (METRIC == 0) ? INFINITY : -INFINITY;  // best distance / similarity so far
    float bs = INFINITY;                              // its pre-sqrt value (euclid only)
    int bi = 0;

    float lse_m = -__builtin_inff(), lse_l = 0.f, ts = 0.f;
    const int64_t tgt = (a.lse_out && a.target && row_ok) ? a.target[row] : (int64_t)-1;
    const int nt = a.n_tiles;
#pragma unroll 1
    for (int ct = 0; ct < nt; ++ct) {
        const int buf = ct & 1;
        VQ_STAMP(0);
        __syncthreads();  // tile ct is in LDS for every wave; everyone is done with the other buffer
        VQ_STAMP(1);

        const char *tile = smem + buf * TILE_B;
        const f32x4 *ap = (const f32x4 *)tile + (hi * 32 + j);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bool more = ct + 1 < nt;
        mfma_sweep_tile<DT>(ap, xr, acc, (const char *)a.packed + (size_t)(more ? ct + 1 : ct) * TILE_B + piece_off,
                            smem + (buf ^ 1) * TILE_B + piece_off, more ? my_pieces : 0);
        VQ_STAMP(2);
        argmin_tile<METRIC>(acc, (const float *)tile + 32 * DT, x2, ct * 32, hi, a.C, bd, bs, bi);
        VQ_STAMP(3);
        if (a.scores_out) {   // cold path: materialise dist[row, code] (vqp.py:741-743) for top-k / sampling / CE / diversity
            const float *y2s = (const float *)tile + 32 * DT;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int code = ct * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
                float v = acc[e];
                if (METRIC == 0) {
                    const float t = x2 + y2s[8 * (e >> 2) + 4 * hi + (e & 3)];
                    v = -sqrtf(fmaxf(__builtin_fmaf(-2.0f, v, t), 1e-8f));
                }
                if (row_ok && code < a.C) a.scores_out[row * a.lds + code] = v;
            }
        }
        if (a.lse_out) {      // cold path: online log-sum-exp of the same dist values, lane-local over this lane's 16 codes per tile
            const float *y2s = (const float *)tile + 32 * DT;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int code = ct * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
                float v = acc[e];
                if (METRIC == 0) {
                    const float t = x2 + y2s[8 * (e >> 2) + 4 * hi + (e & 3)];
                    v = -sqrtf(fmaxf(__builtin_fmaf(-2.0f, v, t), 1e-8f));
                }
                if (code < a.C) {
                    const float nm = fmaxf(lse_m, v);
                    lse_l = lse_l * expf(lse_m - nm) + expf(v - nm);     // (-inf - finite -> exp = 0 on the first code)
                    lse_m = nm;
                    if ((int64_t)code == tgt) ts = v;
                }
            }
        }
    }

    // ---- merge the two half-waves (same row, disjoint code subsets) ------------------------------
    {
        const float od = __shfl_xor(bd, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        const bool take = (METRIC == 0) ? ((od < bd) || (od == bd && oi < bi)) : ((od > bd) || (od == bd && oi < bi));
        bd = take ? od : bd;
        bi = take ? oi : bi;
    }

    if (a.lse_out) {          // merge the half-waves' (max, sum) pairs; the target's score sits in exactly one of them
        const float om = __shfl_xor(lse_m, 32, 64), ol = __shfl_xor(lse_l, 32, 64), ot = __shfl_xor(ts, 32, 64);
        const float M = fmaxf(lse_m, om);
        const float Lsum = lse_l * expf(lse_m - M) + ol * expf(om - M);
        if (row_ok && hi == 0) {
            a.lse_out[row] = M + logf(Lsum);
            if (a.tscore_out) a.tscore_out[row] = a.target ? (tgt < 0 ? 0.f : (tgt >= a.C ? __builtin_nanf("") : ts + ot)) : ((METRIC == 0) ? -bd : bd);   // target >= C: NaN (F.cross_entropy raises there; no silent 0)
        }
    }
    if (row_ok && hi == 0) {
        a.idx_out[row] = (int64_t)bi;
        if (a.best_out) a.best_out[row] = bd;
        if (a.rnorm_out) a.rnorm_out[row] = (METRIC == 0) ? x2 : nrm;
    }

    // ---- commitment-loss partial: sum over the wave's rows of sum_d (q - x)^2, x still in registers (before the q copy: the
    //      rows' registers are dead afterwards and the copy's staging takes their place -- D = 512 spilled 600 registers otherwise) ----
    if (DT == 512 && a.sqerr_partial) {
        // D = 512: the rows fill 256 registers; keeping them alive past the sweep for this sum cost 470 - 660 spilled registers.
        // x and the winning code rows are re-read instead, a row at a time across the wave (coalesced; the fallback kernel's loss only).
        const int64_t wrow0 = (int64_t)blockIdx.x * VQHIP_ASSIGN_ROWS_PER_BLOCK + wave * 32;
        const bool qb = a.q_bf16 != 0;
        double ds = 0.0;
        if (METRIC == 0 && a.x_vec) {
            // whole rows per wave instruction (lane l: elements 4 l .. 4 l + 3 of both 256-column halves), four rows in flight
#pragma unroll 1
            for (int r0 = 0; r0 < 32; r0 += 4) {
                float xv[4][8], gv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t rr = wrow0 + r0 + u;
                    const int64_t rc = rr < a.N ? rr : a.N - 1;
                    const int c = __builtin_amdgcn_readlane(bi, r0 + u);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int d = h * 256 + lane * 4;
                        if (XBF16) {
                            const uint2 w = *(const uint2 *)((const unsigned short *)a.x + rc * a.ldx + d);
                            xv[u][4 * h + 0] = __uint_as_float(w.x << 16); xv[u][4 * h + 1] = __uint_as_float(w.x & 0xffff0000u);
                            xv[u][4 * h + 2] = __uint_as_float(w.y << 16); xv[u][4 * h + 3] = __uint_as_float(w.y & 0xffff0000u);
                        } else {
                            const f32x4 w = *(const f32x4 *)((const float *)a.x + rc * a.ldx + d);
                            xv[u][4 * h + 0] = w.x; xv[u][4 * h + 1] = w.y; xv[u][4 * h + 2] = w.z; xv[u][4 * h + 3] = w.w;
                        }
                        if (qb) {
                            const uint2 w = *(const uint2 *)(a.embed_bf16 + (size_t)c * DT + d);
                            gv[u][4 * h + 0] = __uint_as_float(w.x << 16); gv[u][4 * h + 1] = __uint_as_float(w.x & 0xffff0000u);
                            gv[u][4 * h + 2] = __uint_as_float(w.y << 16); gv[u][4 * h + 3] = __uint_as_float(w.y & 0xffff0000u);
                        } else {
                            const f32x4 w = *(const f32x4 *)(a.embed + (size_t)c * DT + d);
                            gv[u][4 * h + 0] = w.x; gv[u][4 * h + 1] = w.y; gv[u][4 * h + 2] = w.z; gv[u][4 * h + 3] = w.w;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t rr = wrow0 + r0 + u;
                    const bool counted = rr < a.N && (!a.row_mask || a.row_mask[rr] != 0);
                    float ls = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float df = gv[u][k] - xv[u][k]; ls += df * df; }
                    ds += counted ? (double)ls : 0.0;
                }
            }
        } else
#pragma unroll 1
        for (int r = 0; r < 32; ++r) {
            const int64_t rr = wrow0 + r;
            if (rr >= a.N) break;
            if (a.row_mask && a.row_mask[rr] == 0) continue;
            const int c = __builtin_amdgcn_readlane(bi, r);
            const float nr = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(nrm), r));
            float ls = 0.f;
            for (int d = lane; d < a.D; d += 64) {
                float g = a.embed[(size_t)c * a.D + d];
                if (qb) g = round_to_bf16(g);
                float xv = load_elem<XBF16>(a.x, rr * a.ldx + d);
                if (METRIC == 1 && !a.skip_norm) {      // the l2norm the prologue applied to the registers (vqp.py:37-38)
                    xv = xv / nr;
                    if (XBF16) xv = round_to_bf16(xv);
                }
                const float df = g - xv;
                ls += df * df;
            }
            ds += (double)ls;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ds += __shfl_xor(ds, o, 64);
        __syncthreads();
        double *red = (double *)smem;
        if (lane == 0) red[wave] = ds;
        __syncthreads();
        if (tid == 0) a.sqerr_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    if (DT < 512 && a.sqerr_partial) {
        const float *er = a.embed + (size_t)bi * a.D;
        const unsigned short *erb = a.embed_bf16 + (size_t)bi * a.D;
        const bool qb = a.q_bf16 != 0;
        float lsum = 0.f;
        f32x2 lsum2 = {0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            const int k0 = 8 * m + 4 * hi;
            swap32(xr[4 * m + 0], xr[4 * m + 1]);   // this group back to the load layout (the swap is an involution)
            swap32(xr[4 * m + 2], xr[4 * m + 3]);
            float g[4];
            if (a.x_vec && qb) {        // 4 pre-rounded bf16 values in one 8-byte load
                const uint2 w = *(const uint2 *)(erb + k0);
                g[0] = __uint_as_float(w.x << 16); g[1] = __uint_as_float(w.x & 0xffff0000u);
                g[2] = __uint_as_float(w.y << 16); g[3] = __uint_as_float(w.y & 0xffff0000u);
            } else if (a.x_vec) {       // D == DT: embed rows are 16-byte aligned whenever embed is
                const f32x4 w = *(const f32x4 *)(er + k0);
                g[0] = w.x; g[1] = w.y; g[2] = w.z; g[3] = w.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    g[r] = (k0 + r < a.D) ? er[k0 + r] : 0.f;
                    if (qb) g[r] = round_to_bf16(g[r]);
                }
            }
            if (DT <= 256) {
                const f32x2 d01 = f32x2{g[0], g[1]} - f32x2{xr[4 * m + 0], xr[4 * m + 1]};
                const f32x2 d23 = f32x2{g[2], g[3]} - f32x2{xr[4 * m + 2], xr[4 * m + 3]};
                lsum2 = __builtin_elementwise_fma(d01, d01, lsum2);
                lsum2 = __builtin_elementwise_fma(d23, d23, lsum2);
            } else {   // DT = 512 (x spread over VGPRs + AGPRs): the packed form measured ~1e-3 relative error on the GPU
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float df = g[r] - xr[4 * m + r];
                    lsum += df * df;
                }
            }
            // at most 8 code-row loads in flight: requested all at once (NG = 64 at D = 512: 256 registers next to the 256 of the
            // rows) they cost the D = 512 fp32 kernel 470 spilled registers
            if ((m & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        lsum += lsum2[0] + lsum2[1];
        const bool counted = row_ok && (!a.row_mask || a.row_mask[row] != 0);
        double ds = counted ? (double)lsum : 0.0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ds += __shfl_xor(ds, o, 64);
        // all DMA traffic has drained (last loop iteration waited); reuse LDS for the 4 partials
        __syncthreads();
        double *red = (double *)smem;
        if (lane == 0) red[wave] = ds;
        __syncthreads();
        if (tid == 0) a.sqerr_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
    // ---- q = embed[idx]: row-cooperative copy.  For each of the wave's 32 rows all 64 lanes move the
    //      winning code row L2 -> HBM as whole contiguous rows (per-lane-row stores of 8..16 bytes were
    //      measured at 1.7x write amplification: profiles/r1_first).  8 rows in flight per wave. ------
    if (a.q_out) {
        const int64_t wrow0 = (int64_t)blockIdx.x * VQHIP_ASSIGN_ROWS_PER_BLOCK + wave * 32;
        const bool qb = a.q_bf16 != 0;
        // LPR lanes move one row (4 elements each; 64 lanes x 2 passes at D = 512), RPI rows per wave instruction, up to 8
        // instructions in flight.  (One row per instruction with the lanes past the row's end masked off -- the first version --
        // made hipcc spill 1 200 - 1 800 registers at D = 128: every guarded load became a divergent region of its own.)
        constexpr int LPR = DT >= 256 ? 64 : DT / 4;
        constexpr int RPI = 64 / LPR;
        constexpr int NH = (DT + 255) / 256;
        constexpr int NINS = 32 / RPI;                  // wave instructions for the wave's 32 rows
        constexpr int FL = DT > 256 ? 4 : (NINS < 8 ? NINS : 8);   // in flight
        const int sub = lane / LPR, lc4 = (lane % LPR) * 4;
        if (a.q_vec && a.x_vec && qb) {     // bf16 out: verbatim copy of the pre-rounded rows, 8 bytes per lane
#pragma unroll
            for (int i0 = 0; i0 < NINS; i0 += FL) {
                uint2 g[FL][NH];
#pragma unroll
                for (int u = 0; u < FL; ++u) {
                    const int rl = (i0 + u) * RPI + sub;
                    const int c = RPI == 1 ? __builtin_amdgcn_readlane(bi, rl) : __shfl(bi, rl, 64);
                    const unsigned short *er = a.embed_bf16 + (size_t)c * DT + lc4;
#pragma unroll
                    for (int h = 0; h < NH; ++h) g[u][h] = *(const uint2 *)(er + h * 256);
                }
#pragma unroll
                for (int u = 0; u < FL; ++u) {
                    const int64_t rr = wrow0 + (i0 + u) * RPI + sub;
                    if (rr < a.N) {
#pragma unroll
                        for (int h = 0; h < NH; ++h) *(uint2 *)((unsigned short *)a.q_out + rr * a.ldq + lc4 + h * 256) = g[u][h];
                    }
                }
            }
        } else if (a.q_vec && a.x_vec) {
#pragma unroll
            for (int i0 = 0; i0 < NINS; i0 += FL) {
                f32x4 g[FL][NH];
#pragma unroll
                for (int u = 0; u < FL; ++u) {
                    const int rl = (i0 + u) * RPI + sub;
                    const int c = RPI == 1 ? __builtin_amdgcn_readlane(bi, rl) : __shfl(bi, rl, 64);
                    const float *er = a.embed + (size_t)c * DT + lc4;
#pragma unroll
                    for (int h = 0; h < NH; ++h) g[u][h] = *(const f32x4 *)(er + h * 256);
                }
#pragma unroll
                for (int u = 0; u < FL; ++u) {
                    const int64_t rr = wrow0 + (i0 + u) * RPI + sub;
                    if (rr < a.N) {
#pragma unroll
                        for (int h = 0; h < NH; ++h) *(f32x4 *)((float *)a.q_out + rr * a.ldq + lc4 + h * 256) = g[u][h];
                    }
                }
            }
        } else {
            for (int r = 0; r < 32; ++r) {
                const int64_t rr = wrow0 + r;
                if (rr >= a.N) break;
                const int c = __builtin_amdgcn_readlane(bi, r);
                const float *er = a.embed + (size_t)c * a.D;
                for (int d = lane; d < a.D; d += 64) {
                    if (qb) ((unsigned short *)a.q_out)[rr * a.ldq + d] = f32_to_bf16_rne(er[d]);
                    else ((float *)a.q_out)[rr * a.ldq + d] = er[d];
                }
            }
        }
    }

}

extern "C" int64_t vqhip_assign_blocks(int64_t N)
{
    return N <= 0 ? 0 : (N + VQHIP_ASSIGN_ROWS_PER_BLOCK - 1) / VQHIP_ASSIGN_ROWS_PER_BLOCK;
}

template <int DT, bool XBF16, int METRIC>
static int launch_assign(const AssignArgs &a, hipStream_t st)
{
    constexpr int TILE_B = (32 * DT + 256) * 4;
    constexpr int SMEM = 2 * TILE_B;
    static VqAttrOnce once;   // per instantiation and device
    if (int rc = vq_set_max_smem(once, (const void *)vq_assign_kernel<DT, XBF16, METRIC>, SMEM, "vq_assign_kernel")) return rc;
    const int64_t blocks = vqhip_assign_blocks(a.N);
    hipLaunchKernelGGL((vq_assign_kernel<DT, XBF16, METRIC>), dim3((unsigned)blocks, (unsigned)(a.heads > 1 ? a.heads : 1)), dim3(256), SMEM, st, a);
    return launch_status("vq_assign_kernel");
}

template <int DT>
static int dispatch_assign(const AssignArgs &a, int x_dtype, int metric, hipStream_t st)
{
    if (x_dtype == VQHIP_BF16)
        return metric ? launch_assign<DT, true, 1>(a, st) : launch_assign<DT, true, 0>(a, st);
    return metric ? launch_assign<DT, false, 1>(a, st) : launch_assign<DT, false, 0>(a, st);
}

// ------------------------------------------------------------------------------------------------
// exact top-K codes per row (vqp.py:137-138: `logits.topk(topk)` of dist = -cdist or the cosine similarity) without the N x C
// `dist` tensor: the sweep of vq_assign_kernel with a per-lane sorted list of the K best (score, code) instead of the running
// argmin.  Scores are the reference's values (correctly rounded sqrt per element), order = (score descending, code ascending).
// Cold path (forward(topk=), ResidualVQ beam search): K <= 8, D in {32, 64, 128, 256, 512}, vector-aligned rows.
// ------------------------------------------------------------------------------------------------
#define VQ_TOPK_MAX 8
struct TopkArgs {
    const void *x;
    int64_t N;
    int64_t ldx;
    const float *packed;
    int C;
    int n_tiles;
    int K;
    int skip_norm;
    int64_t *idx_out;   // [N, K]
    float *val_out;     // nullable [N, K]
};

__device__ __forceinline__ bool topk_better(float v, int c, float lv, int lc) { return v > lv || (v == lv && c < lc); }

// bubble one item down a list sorted by (score descending, code ascending): K compare-swaps, branch-free; entries past K stay unused
__device__ __noinline__ void topk_insert(float (&lv)[VQ_TOPK_MAX], int (&lc)[VQ_TOPK_MAX], int K, float v, int c)
{
#pragma unroll
    for (int k = 0; k < VQ_TOPK_MAX; ++k) {
        const bool b = (k < K) && topk_better(v, c, lv[k], lc[k]);
        const float tv = lv[k]; const int tc = lc[k];
        lv[k] = b ? v : tv; lc[k] = b ? c : tc;
        v = b ? tv : v; c = b ? tc : c;
    }
}

template <int DT, bool XBF16, int METRIC>
__global__ void __launch_bounds__(256, (DT <= 256 ? 2 : 1)) vq_topk_kernel(const TopkArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_F = 32 * DT + 256;
    constexpr int TILE_B = TILE_F * 4;
    constexpr int NCHUNK = TILE_B / 1024;
    constexpr int NG = DT / 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int hi = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * VQHIP_ASSIGN_ROWS_PER_BLOCK + wave * 32 + j;
    const bool row_ok = row < a.N;
    const int64_t rowc = row_ok ? row : (a.N - 1);
    const int my_pieces = (NCHUNK - wave + 3) / 4;
    const int piece_off = wave * 1024 + lane * 16;
    for (int k = 0; k < my_pieces; ++k)
        *(f32x4 *)(smem + piece_off + k * 4096) = *(const f32x4 *)((const char *)a.packed + piece_off + (size_t)k * 4096);

    float xr[DT / 2];   // load layout, as in vq_assign_kernel
    if (XBF16) {
        const uint2 *p = (const uint2 *)((const unsigned short *)a.x + rowc * a.ldx + 4 * hi);
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            const uint2 w = p[m * 2];
            xr[4 * m + 0] = __uint_as_float(w.x << 16); xr[4 * m + 1] = __uint_as_float(w.x & 0xffff0000u);
            xr[4 * m + 2] = __uint_as_float(w.y << 16); xr[4 * m + 3] = __uint_as_float(w.y & 0xffff0000u);
        }
    } else {
        const f32x4 *p = (const f32x4 *)((const float *)a.x + rowc * a.ldx + 4 * hi);
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            const f32x4 w = p[m * 2];
            xr[4 * m + 0] = w.x; xr[4 * m + 1] = w.y; xr[4 * m + 2] = w.z; xr[4 * m + 3] = w.w;
        }
    }
    const float x2 = x2_aten_order<DT>(xr, j);
    if (METRIC == 1 && !a.skip_norm) {
        float nrm = sqrtf(x2);
        if (XBF16) nrm = round_to_bf16(nrm);
        nrm = fmaxf(nrm, XBF16 ? round_to_bf16(1e-6f) : 1e-6f);
#pragma unroll
        for (int q = 0; q < DT / 2; ++q) { const float v = xr[q] / nrm; xr[q] = XBF16 ? round_to_bf16(v) : v; }
    }
#pragma unroll
    for (int m = 0; m < NG; ++m) { swap32(xr[4 * m + 0], xr[4 * m + 1]); swap32(xr[4 * m + 2], xr[4 * m + 3]); }

    float lv[VQ_TOPK_MAX];
    int lc[VQ_TOPK_MAX];
#pragma unroll
    for (int k = 0; k < VQ_TOPK_MAX; ++k) { lv[k] = -INFINITY; lc[k] = 0x7fffffff; }
    const int K = a.K;

    const int nt = a.n_tiles;
    for (int ct = 0; ct < nt; ++ct) {
        const int buf = ct & 1;
        __syncthreads();
        const char *tile = smem + buf * TILE_B;
        const f32x4 *ap = (const f32x4 *)tile + (hi * 32 + j);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const bool more = ct + 1 < nt;
        mfma_sweep_tile<DT>(ap, xr, acc, (const char *)a.packed + (size_t)(more ? ct + 1 : ct) * TILE_B + piece_off,
                            smem + (buf ^ 1) * TILE_B + piece_off, more ? my_pieces : 0);
        // the 16 scores of this lane go through LDS-free scratch-free insertion one by one; the (cold-path) loop over the
        // elements is a real loop -- the scores are staged in this lane's slice of a small LDS array so that it can be
        // indexed at run time (a register array cannot)
        const float *y2s = (const float *)tile + 32 * DT;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int code = ct * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
            float t = acc[e];
            if (METRIC == 0) t = -sqrtf(fmaxf(__builtin_fmaf(-2.0f, t, x2 + y2s[8 * (e >> 2) + 4 * hi + (e & 3)]), 1e-8f));
            acc[e] = (code < a.C) ? t : -INFINITY;
        }
        float vmax = acc[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) vmax = fmaxf(vmax, acc[e]);
        // skip the insertions when no lane can improve its list (a tie would need a lower code: tiles ascend, so never)
        float worst = lv[0];
#pragma unroll
        for (int k = 1; k < VQ_TOPK_MAX; ++k) worst = (k == K - 1) ? lv[k] : worst;
        if (__any(vmax > worst)) {
            float *sv = (float *)(smem + 2 * TILE_B) + tid * 16;
#pragma unroll
            for (int e = 0; e < 16; ++e) sv[e] = acc[e];
#pragma unroll 1
            for (int e = 0; e < 16; ++e)
                topk_insert(lv, lc, K, sv[e], ct * 32 + 8 * (e >> 2) + 4 * hi + (e & 3));
        }
    }
    // merge the two half-waves (same row, disjoint code subsets): the partner's list goes into this lane's
    float pv[VQ_TOPK_MAX];
    int pc[VQ_TOPK_MAX];
#pragma unroll
    for (int k = 0; k < VQ_TOPK_MAX; ++k) { pv[k] = __shfl_xor(lv[k], 32, 64); pc[k] = __shfl_xor(lc[k], 32, 64); }
#pragma unroll 1
    for (int k = 0; k < VQ_TOPK_MAX; ++k) {
        float v = pv[0]; int c = pc[0];
#pragma unroll
        for (int q = 1; q < VQ_TOPK_MAX; ++q) { v = (q == k) ? pv[q] : v; c = (q == k) ? pc[q] : c; }
        topk_insert(lv, lc, K, v, c);
    }
    if (row_ok && hi == 0) {
#pragma unroll
        for (int k = 0; k < VQ_TOPK_MAX; ++k)
            if (k < K) {
                a.idx_out[row * K + k] = (int64_t)lc[k];
                if (a.val_out) a.val_out[row * K + k] = lv[k];
            }
    }
}

template <int DT>
static int dispatch_topk(const TopkArgs &a, int x_dtype, int metric, hipStream_t st)
{
    constexpr int SMEM = 2 * (32 * DT + 256) * 4 + 256 * 16 * 4;   // two tiles + the per-lane score staging of the insertion loop
    const unsigned blocks = (unsigned)vqhip_assign_blocks(a.N);
#define VQ_TK(B, M)                                                                                                        \
    do {                                                                                                                   \
        static VqAttrOnce once;                                                                                            \
        if (int rc = vq_set_max_smem(once, (const void *)vq_topk_kernel<DT, B, M>, SMEM, "vq_topk_kernel")) return rc;     \
        hipLaunchKernelGGL((vq_topk_kernel<DT, B, M>), dim3(blocks), dim3(256), SMEM, st, a);                              \
    } while (0)
    if (metric == VQHIP_EUCLID) { if (x_dtype == VQHIP_BF16) VQ_TK(true, 0); else VQ_TK(false, 0); }
    else                        { if (x_dtype == VQHIP_BF16) VQ_TK(true, 1); else VQ_TK(false, 1); }
#undef VQ_TK
    return launch_status("vq_topk_kernel");
}

extern "C" int vqhip_topk(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed, int C, int metric, int K,
                          int64_t *idx_out, float *val_out, void *stream)
{
    if (!x || !packed || !idx_out || N < 0 || C <= 0) VQ_FAIL(VQHIP_EINVAL, "topk: bad argument");
    if (K < 1 || K > VQ_TOPK_MAX || K > C) VQ_FAIL(VQHIP_EINVAL, "topk: K=%d outside 1..min(%d, C)", K, VQ_TOPK_MAX);
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "topk: unknown dtype %d", x_dtype);
    if (metric != VQHIP_EUCLID && metric != VQHIP_COSINE && metric != VQHIP_COSINE_PRENORM) VQ_FAIL(VQHIP_EINVAL, "topk: unknown metric %d", metric);
    if (D != 32 && D != 64 && D != 128 && D != 256 && D != 512) VQ_FAIL(VQHIP_EDIM, "topk: D=%d unsupported (32, 64, 128, 256, 512)", D);
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    if ((((uintptr_t)x) % (4 * es)) || ((ldx * es) % (4 * es)) || ldx < D) VQ_FAIL(VQHIP_EALIGN, "topk: rows must be aligned to 4 elements");
    if (N == 0) return 0;
    TopkArgs a;
    a.x = x; a.N = N; a.ldx = ldx; a.packed = packed; a.C = C; a.n_tiles = (C + 31) / 32; a.K = K;
    a.skip_norm = metric == VQHIP_COSINE_PRENORM; a.idx_out = idx_out; a.val_out = val_out;
    const int m = metric == VQHIP_EUCLID ? VQHIP_EUCLID : VQHIP_COSINE;
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 32: return dispatch_topk<32>(a, x_dtype, m, st);
        case 64: return dispatch_topk<64>(a, x_dtype, m, st);
        case 128: return dispatch_topk<128>(a, x_dtype, m, st);
        case 256: return dispatch_topk<256>(a, x_dtype, m, st);
        default: return dispatch_topk<512>(a, x_dtype, m, st);
    }
}

// thread-per-row exact ATen-order sum of squares (any D <= 512)
template <bool XBF16>
__global__ void __launch_bounds__(256) vq_row_sumsq_kernel(const void *x, int64_t N, int D, int64_t ldx, float *out)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    out[n] = aten_sumsq_seq([&](int e) { return load_elem<XBF16>(x, n * ldx + e); }, D);
}

// Score of ONE given code per row in the reference's arithmetic: the Euclidean distance cdist(x_n, c_idx) (vqp.py:58-62: ATen-order
// norms, x.c as one ascending fp32 FMA chain, (x2 + y2) + (-2 xy), clamp, correctly rounded sqrt) or, for unit-norm rows, the
// cosine similarity x_n . c_idx (vqp.py:741) -- bit for bit what vq_assign_kernel reports as the winner's score.  The screened
// search does not produce scores; a codebook-sharded argmin needs the winner's exact score to merge shards (one chain per row).
template <bool XBF16, int METRIC>
__global__ void __launch_bounds__(128) vq_score_idx_kernel(const void *x, int64_t N, int D, int64_t ldx, const float *embed,
                                                           const float *packed, int DT, const int64_t *idx, float *out)
{
    // One row per lane keeps the reference's sequential k order; the row and its code are staged through LDS 32 elements at a
    // time so that the global reads are 128-byte row segments instead of one strided 16-byte read per lane.
    __shared__ float xs[128][33], es[128][33];
    __shared__ int64_t cs[128];
    const int t = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * 128, n = n0 + t;
    cs[t] = idx[n < N ? n : N - 1];
    __syncthreads();
    float xy = 0.f, ch[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) ch[c] = 0.f;
    float fin = 0.f;                       // elements past the last whole group of 8 (aten_sumsq_seq's tail)
    const int V8 = (D >> 3) << 3, V32 = (D >> 5) << 5;
    for (int k0 = 0; k0 < D; k0 += 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = t + 128 * i, row = f >> 3, k = k0 + 4 * (f & 7);
            const int64_t rn = (n0 + row < N) ? n0 + row : N - 1;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f}, ev = {0.f, 0.f, 0.f, 0.f};
            if (k < D) {
                if (XBF16) {
                    const uint2 w = *(const uint2 *)((const unsigned short *)x + rn * ldx + k);
                    xv.x = __uint_as_float(w.x << 16); xv.y = __uint_as_float(w.x & 0xffff0000u);
                    xv.z = __uint_as_float(w.y << 16); xv.w = __uint_as_float(w.y & 0xffff0000u);
                } else {
                    xv = *(const f32x4 *)((const float *)x + rn * ldx + k);
                }
                ev = *(const f32x4 *)(embed + (size_t)cs[row] * D + k);
            }
            float *px = &xs[row][4 * (f & 7)], *pe = &es[row][4 * (f & 7)];
            px[0] = xv.x; px[1] = xv.y; px[2] = xv.z; px[3] = xv.w;
            pe[0] = ev.x; pe[1] = ev.y; pe[2] = ev.z; pe[3] = ev.w;
        }
        __syncthreads();
        const int lim = (D - k0 < 32) ? D - k0 : 32;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float xv = xs[t][j];
            if (j < lim) xy = __builtin_fmaf(xv, es[t][j], xy);
            if (METRIC == 0) {
                const int e = k0 + j;                 // chain assignment of aten_sumsq_seq
                const float sq = xv * xv;
                if (e < V32) ch[j] += sq;
                else if (e < V8) ch[j & 7] += sq;
                else if (e < D) fin += sq;
            }
        }
        __syncthreads();
    }
    if (n >= N) return;
    if (METRIC == 0) {
#pragma unroll
        for (int l = 0; l < 8; ++l) fin += ((ch[l] + ch[8 + l]) + ch[16 + l]) + ch[24 + l];
        const int64_t c = cs[t];
        const float y2 = packed[(size_t)(c >> 5) * (32 * DT + 256) + 32 * DT + (c & 31)];
        out[n] = sqrtf(fmaxf(__builtin_fmaf(-2.f, xy, fin + y2), 1e-8f));
    } else {
        out[n] = xy;
    }
}

extern "C" int vqhip_score_indices(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed, const float *embed,
                                   int C, int metric, const int64_t *idx, float *out, void *stream)
{
    if (!x || !packed || !embed || !idx || !out || N < 0 || C <= 0) VQ_FAIL(VQHIP_EINVAL, "score_indices: bad argument");
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "score_indices: unknown dtype %d", x_dtype);
    if (metric != VQHIP_EUCLID && metric != VQHIP_COSINE_PRENORM) VQ_FAIL(VQHIP_EINVAL, "score_indices: metric %d (VQHIP_EUCLID or VQHIP_COSINE_PRENORM)", metric);
    const int DT = pick_dt(D);
    if (DT == 0 || (D & 3)) VQ_FAIL(VQHIP_EDIM, "score_indices: D=%d unsupported (multiple of 4, <= 512)", D);
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    if ((((uintptr_t)x) % (4 * es)) || ((ldx * es) % (4 * es)) || (((uintptr_t)embed) & 15)) VQ_FAIL(VQHIP_EALIGN, "score_indices: rows must be aligned to 4 elements");
    if (N == 0) return 0;
    const unsigned blocks = (unsigned)((N + 127) / 128);
    hipStream_t st = (hipStream_t)stream;
#define VQ_SI(B, M) hipLaunchKernelGGL((vq_score_idx_kernel<B, M>), dim3(blocks), dim3(128), 0, st, x, N, D, ldx, embed, packed, DT, idx, out)
    if (metric == VQHIP_EUCLID) { if (x_dtype == VQHIP_BF16) VQ_SI(true, 0); else VQ_SI(false, 0); }
    else                        { if (x_dtype == VQHIP_BF16) VQ_SI(true, 1); else VQ_SI(false, 1); }
#undef VQ_SI
    return launch_status("vq_score_idx_kernel");
}

// ------------------------------------------------------------------------------------------------
// A separate codebook PER ROW (QINCo's implicit neural codebook: the reference hands Codebook.forward a codebook_transform_fn,
// vqp.py:729-738, whose output is [h, b, n, c, d]): idx[n] = argmin_c ||x_n - e_{n,c} + 1e-6||_2 (F.pairwise_distance's eps)
// or, for the cosine metric, argmax_c x_n . e_{n,c}; first extremum in ascending c, like ATen's argmax of -dist.  There is no
// codebook to keep resident: the kernel streams N * C * D floats once (HBM bound).  One wave per row, 4 codes per step so that
// the cross-lane sums of one step overlap; lanes take the features in 4-element slices (D % 4 == 0) or one by one.
// ------------------------------------------------------------------------------------------------
template <int METRIC, bool VEC>
__global__ void __launch_bounds__(256) vq_rowwise_kernel(const float *x, int64_t N, int D, int64_t ldx, const float *codes, int C, int64_t *idx_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float *xr = x + n * ldx;
    const float *er = codes + (size_t)n * C * D;
    float best = METRIC == 0 ? INFINITY : -INFINITY;
    int bi = 0;
    for (int c0 = 0; c0 < C; c0 += 4) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
            for (int d = lane * 4; d < D; d += 256) {
                const f32x4 xv = *(const f32x4 *)(xr + d);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 ev = *(const f32x4 *)(er + (size_t)min(c0 + u, C - 1) * D + d);
                    if (METRIC == 0) {
                        const float t0 = (xv.x - ev.x) + 1e-6f, t1 = (xv.y - ev.y) + 1e-6f, t2 = (xv.z - ev.z) + 1e-6f, t3 = (xv.w - ev.w) + 1e-6f;
                        s[u] = __builtin_fmaf(t0, t0, s[u]); s[u] = __builtin_fmaf(t1, t1, s[u]);
                        s[u] = __builtin_fmaf(t2, t2, s[u]); s[u] = __builtin_fmaf(t3, t3, s[u]);
                    } else {
                        s[u] = __builtin_fmaf(xv.x, ev.x, s[u]); s[u] = __builtin_fmaf(xv.y, ev.y, s[u]);
                        s[u] = __builtin_fmaf(xv.z, ev.z, s[u]); s[u] = __builtin_fmaf(xv.w, ev.w, s[u]);
                    }
                }
            }
        } else {
            for (int d = lane; d < D; d += 64) {
                const float xv = xr[d];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ev = er[(size_t)min(c0 + u, C - 1) * D + d];
                    if (METRIC == 0) { const float t = (xv - ev) + 1e-6f; s[u] = __builtin_fmaf(t, t, s[u]); }
                    else s[u] = __builtin_fmaf(xv, ev, s[u]);
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += __shfl_xor(s[u], o, 64);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u < C) {
                const float v = METRIC == 0 ? sqrtf(s[u]) : s[u];
                if (METRIC == 0 ? (v < best) : (v > best)) { best = v; bi = c0 + u; }
            }
        }
    }
    if (lane == 0) idx_out[n] = (int64_t)bi;
}

extern "C" int vqhip_assign_rowwise(const float *x, int64_t N, int D, int64_t ldx, const float *codes, int C, int metric,
                                    int64_t *idx_out, void *stream)
{
    if (!x || !codes || !idx_out || N < 0 || C <= 0 || D < 1) VQ_FAIL(VQHIP_EINVAL, "assign_rowwise: bad argument");
    if (metric != VQHIP_EUCLID && metric != VQHIP_COSINE_PRENORM) VQ_FAIL(VQHIP_EINVAL, "assign_rowwise: metric must be VQHIP_EUCLID or VQHIP_COSINE_PRENORM");
    if (N == 0) return 0;
    const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x) % 16 == 0) && (((uintptr_t)codes) % 16 == 0);
    const unsigned blocks = (unsigned)((N + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (metric == VQHIP_EUCLID) {
        if (vec) hipLaunchKernelGGL((vq_rowwise_kernel<0, true>), dim3(blocks), dim3(256), 0, st, x, N, D, ldx, codes, C, idx_out);
        else hipLaunchKernelGGL((vq_rowwise_kernel<0, false>), dim3(blocks), dim3(256), 0, st, x, N, D, ldx, codes, C, idx_out);
    } else {
        if (vec) hipLaunchKernelGGL((vq_rowwise_kernel<1, true>), dim3(blocks), dim3(256), 0, st, x, N, D, ldx, codes, C, idx_out);
        else hipLaunchKernelGGL((vq_rowwise_kernel<1, false>), dim3(blocks), dim3(256), 0, st, x, N, D, ldx, codes, C, idx_out);
    }
    return launch_status("vq_rowwise_kernel");
}

extern "C" int vqhip_row_sumsq(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, float *out, void *stream)
{
    if (!x || !out || N < 0) VQ_FAIL(VQHIP_EINVAL, "row_sumsq: bad argument");
    if (D < 1 || D > VQ_WIDE_MAX_D) VQ_FAIL(VQHIP_EDIM, "row_sumsq: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (N == 0) return 0;
    if (x_dtype != VQHIP_BF16 && x_dtype != VQHIP_F32) VQ_FAIL(VQHIP_EINVAL, "row_sumsq: unknown dtype %d", x_dtype);
    if (vq_is_wide(D)) return vq_wide_row_sumsq(x, x_dtype, N, D, ldx, out, stream);
    const unsigned blocks = (unsigned)((N + 255) / 256);
    if (x_dtype == VQHIP_BF16)
        hipLaunchKernelGGL(vq_row_sumsq_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, D, ldx, out);
    else if (x_dtype == VQHIP_F32)
        hipLaunchKernelGGL(vq_row_sumsq_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, N, D, ldx, out);
    else
        VQ_FAIL(VQHIP_EINVAL, "row_sumsq: unknown dtype %d", x_dtype);
    return launch_status("vq_row_sumsq_kernel");
}

static int assign_impl(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                       const float *packed, const float *embed, int C, int metric,
                       int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq,
                       float *best_out, float *rnorm_out, double *sqerr_partial,
                       const uint8_t *row_mask, float *scores_out, int64_t lds, void *stream, float *lse_out, float *tscore_out,
                       const int64_t *target, int heads, int64_t x_hstride, int64_t q_hstride);

extern "C" int vqhip_assign(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                            const float *packed, const float *embed, int C, int metric,
                            int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq,
                            float *best_out, float *rnorm_out, double *sqerr_partial,
                            const uint8_t *row_mask, void *stream)
{
    return assign_impl(x, x_dtype, N, D, ldx, packed, embed, C, metric, idx_out, q_out, q_dtype, ldq, best_out, rnorm_out,
                       sqerr_partial, row_mask, nullptr, 0, stream, nullptr, nullptr, nullptr, 1, 0, 0);
}

// the exact search of H heads in one launch (grid dimension y = head): dims the screened search does not take (e.g. the 16-wide
// heads of RandomProjectionQuantizer).  Layout of the heads' buffers as in vqhip_assign_screened_batched; rnorm_out [H, N].
extern "C" int vqhip_assign_batched(const void *x, int x_dtype, int H, int64_t N, int D, int64_t ldx, int64_t x_hstride,
                                    const float *packed, const float *embed, int C, int metric,
                                    int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq, int64_t q_hstride,
                                    float *rnorm_out, const uint8_t *row_mask, void *stream)
{
    if (H < 1) VQ_FAIL(VQHIP_EINVAL, "assign_batched: H < 1");
    return assign_impl(x, x_dtype, N, D, ldx, packed, embed, C, metric, idx_out, q_out, q_dtype, ldq, nullptr, rnorm_out,
                       nullptr, row_mask, nullptr, 0, stream, nullptr, nullptr, nullptr, H, x_hstride, q_hstride);
}

extern "C" int vqhip_scores(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                            const float *packed, const float *embed, int C, int metric,
                            float *scores_out, int64_t lds, int64_t *idx_out, float *rnorm_out, void *stream)
{
    if (!scores_out || lds < C) VQ_FAIL(VQHIP_EINVAL, "scores: scores_out null or row stride smaller than C");
    return assign_impl(x, x_dtype, N, D, ldx, packed, embed, C, metric, idx_out, nullptr, VQHIP_F32, D, nullptr, rnorm_out,
                       nullptr, nullptr, scores_out, lds, stream, nullptr, nullptr, nullptr, 1, 0, 0);
}

extern "C" int vqhip_scores_lse(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                                const float *packed, const float *embed, int C, int metric, const int64_t *target,
                                float *lse_out, float *tscore_out, int64_t *idx_out, float *rnorm_out, void *stream)
{
    if (!lse_out) VQ_FAIL(VQHIP_EINVAL, "scores_lse: lse_out is null");
    return assign_impl(x, x_dtype, N, D, ldx, packed, embed, C, metric, idx_out, nullptr, VQHIP_F32, D, nullptr, rnorm_out,
                       nullptr, nullptr, nullptr, 0, stream, lse_out, tscore_out, target, 1, 0, 0);
}

static int assign_impl(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                            const float *packed, const float *embed, int C, int metric,
                            int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq,
                            float *best_out, float *rnorm_out, double *sqerr_partial,
                            const uint8_t *row_mask, float *scores_out, int64_t lds, void *stream, float *lse_out, float *tscore_out,
                            const int64_t *target, int heads, int64_t x_hstride, int64_t q_hstride)
{
    if (N < 0 || C <= 0) VQ_FAIL(VQHIP_EINVAL, "assign: N < 0 or C <= 0");
    if (heads > 1 && (best_out || sqerr_partial || scores_out || lse_out))
        VQ_FAIL(VQHIP_EINVAL, "assign: a batched launch has index, q and rnorm outputs only");
    if (N == 0) return 0;
    if (!x || !packed || !embed || !idx_out) VQ_FAIL(VQHIP_EINVAL, "assign: null pointer");
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "assign: unknown x dtype %d", x_dtype);
    if (q_out && q_dtype != VQHIP_F32 && q_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "assign: unknown q dtype %d", q_dtype);
    if (metric < 0 || metric > 2) VQ_FAIL(VQHIP_EINVAL, "assign: unknown metric %d", metric);
    const int DT = pick_dt(D);
    if (D < 1 || (DT == 0 && !vq_is_wide(D))) VQ_FAIL(VQHIP_EDIM, "assign: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (ldx < D || (q_out && ldq < D)) VQ_FAIL(VQHIP_EINVAL, "assign: row stride smaller than D");
    if (((uintptr_t)packed) & 15) VQ_FAIL(VQHIP_EALIGN, "assign: packed must be 16-byte aligned");
    if (vq_is_wide(D)) {        // 512 < D <= 2048: the plain exact kernel of vq_wide.hip (index / q / score / norm / squared-error outputs)
        if (scores_out || lse_out || heads > 1)
            VQ_FAIL(VQHIP_EDIM, "assign: D=%d: score rows, the streaming log-sum-exp and batched heads stop at D = 512", D);
        return vq_wide_assign(x, x_dtype, N, D, ldx, packed, embed, C, metric, idx_out, q_out, q_dtype, ldq, best_out, rnorm_out, sqerr_partial,
                              row_mask, stream);
    }
    if ((D & 31) && !rnorm_out) VQ_FAIL(VQHIP_EINVAL, "assign: D %% 32 != 0 needs rnorm_out (scratch for the exact ||x||^2 pre-pass)");

    hipStream_t st = (hipStream_t)stream;
    if (D & 31) {  // exact ATen-order ||x||^2 for odd D: pre-pass into rnorm_out, the kernel reads it back
        for (int h = 0; h < (heads > 1 ? heads : 1); ++h) {
            int rc = vqhip_row_sumsq((const char *)x + (size_t)h * x_hstride * ((x_dtype == VQHIP_BF16) ? 2 : 4), x_dtype, N, D, ldx,
                                     rnorm_out + (size_t)h * N, stream);
            if (rc) return rc;
        }
    }

    AssignArgs a;
    a.x = x; a.N = N; a.D = D; a.ldx = ldx; a.packed = packed; a.embed = embed; a.C = C;
    a.embed_bf16 = (const unsigned short *)((const char *)packed + packed_bf16_offset(C, D));
    a.n_tiles = (C + 31) / 32;
    a.idx_out = idx_out; a.q_out = q_out; a.q_bf16 = (q_dtype == VQHIP_BF16); a.ldq = ldq;
    a.skip_norm = (metric == VQHIP_COSINE_PRENORM);
#ifdef VQ_TRACE
    a.trace = g_trace;
#endif
    a.scores_out = scores_out; a.lds = lds;
    a.lse_out = lse_out; a.tscore_out = tscore_out; a.target = target;
    a.best_out = best_out; a.rnorm_out = rnorm_out; a.sqerr_partial = sqerr_partial; a.row_mask = row_mask;
    const int xes = (x_dtype == VQHIP_BF16) ? 2 : 4;
    a.heads = heads > 1 ? heads : 1;
    a.hs_x = x_hstride * xes; a.hs_packed = (int64_t)vq_packed_total_bytes(C, D); a.hs_embed = (int64_t)C * D * 4;
    a.hs_idx = N * 8; a.hs_q = q_hstride * ((q_dtype == VQHIP_BF16) ? 2 : 4); a.hs_rnorm = N * 4;
    if (heads > 1 && (((x_hstride * xes) % (4 * xes)) || (q_out && ((q_hstride * ((q_dtype == VQHIP_BF16) ? 2 : 4)) % 16))))
        VQ_FAIL(VQHIP_EALIGN, "assign: heads' rows must stay aligned to 4 elements");
    a.x_vec = (D == DT) && (((uintptr_t)x) % (4 * xes) == 0) && ((ldx * xes) % (4 * xes) == 0) && ((((uintptr_t)embed) & 15) == 0);
    if (q_out) {
        const int qes = a.q_bf16 ? 2 : 4;
        a.q_vec = (D == DT) && (((uintptr_t)q_out) % (4 * qes) == 0) && ((ldq * qes) % (4 * qes) == 0);
    } else {
        a.q_vec = 0;
    }

    switch (DT) {
        case 32: return dispatch_assign<32>(a, x_dtype, metric, st);
        case 64: return dispatch_assign<64>(a, x_dtype, metric, st);
        case 128: return dispatch_assign<128>(a, x_dtype, metric, st);
        case 256: return dispatch_assign<256>(a, x_dtype, metric, st);
        default: return dispatch_assign<512>(a, x_dtype, metric, st);  // metric != 0 -> cosine family
    }
}

// ------------------------------------------------------------------------------------------------
// l2norm of the rows (vqp.py:37-38 applied at :1159), in exactly the arithmetic vq_assign_kernel<.., COSINE> uses in its
// prologue: ||x||^2 in ATen's order, x / max(||x||, 1e-6), for bf16 tensors norm and quotient rounded to bf16 as the
// reference's bf16 ops do.  Lets the cosine metric run through the screened search (which takes unit-norm rows).
// D == DT in {32, 64, 128, 256}, vector-aligned rows.  32 rows per wave, same load layout as the assign kernels.
// ------------------------------------------------------------------------------------------------
template <int DT, bool XBF16>
__global__ void __launch_bounds__(256) vq_l2norm_kernel(const void *x, int64_t N, int64_t ldx, void *out, int64_t ldo)
{
    constexpr int NG = DT / 8;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31;
    const int hi = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + j;
    const bool row_ok = row < N;
    const int64_t rowc = row_ok ? row : (N - 1);
    float xr[DT / 2];
    if (XBF16) {
        const uint2 *p = (const uint2 *)((const unsigned short *)x + rowc * ldx + 4 * hi);
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            const uint2 w = p[m * 2];
            xr[4 * m + 0] = __uint_as_float(w.x << 16);
            xr[4 * m + 1] = __uint_as_float(w.x & 0xffff0000u);
            xr[4 * m + 2] = __uint_as_float(w.y << 16);
            xr[4 * m + 3] = __uint_as_float(w.y & 0xffff0000u);
        }
    } else {
        const f32x4 *p = (const f32x4 *)((const float *)x + rowc * ldx + 4 * hi);
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            const f32x4 w = p[m * 2];
            xr[4 * m + 0] = w.x; xr[4 * m + 1] = w.y; xr[4 * m + 2] = w.z; xr[4 * m + 3] = w.w;
        }
    }
    const float x2 = x2_aten_order<DT>(xr, j);
    float nrm = sqrtf(x2);
    if (XBF16) nrm = round_to_bf16(nrm);
    nrm = fmaxf(nrm, XBF16 ? round_to_bf16(1e-6f) : 1e-6f);
    if (!row_ok) return;
#pragma unroll
    for (int m = 0; m < NG; ++m) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = xr[4 * m + r] / nrm;
            if (XBF16) v[r] = round_to_bf16(v[r]);
        }
        if (XBF16) {
            uint2 w;
            w.x = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
            w.y = (__float_as_uint(v[2]) >> 16) | (__float_as_uint(v[3]) & 0xffff0000u);
            *(uint2 *)((unsigned short *)out + row * ldo + 8 * m + 4 * hi) = w;
        } else {
            *(f32x4 *)((float *)out + row * ldo + 8 * m + 4 * hi) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

// The same arithmetic with coalesced accesses: 16 lanes per row, every lane owns 16-byte pieces (elements 64 k + 4 l + e), so a wave
// instruction touches four rows' contiguous 256 bytes (the kernel above: 8-byte pieces of 32 different rows per instruction -- 2.9 TB/s
// on 2^20 x 256 bf16 rows).  ATen's summation order is kept by walking its 32 chains (elements i = res + 32 t in increasing t, res =
// 8 (m & 3) + 4 hi + r in x2_aten_order's terms) across the two lanes that hold them: chain `res` lives on lane res / 4 (e = res & 3)
// for even t and on lane res / 4 + 8 for odd t; then the same p / f_lo / f_hi combination.  D = 32 T, T in {1, 2, 4, 8, 16}.
template <int NK, bool XBF16>
__global__ void __launch_bounds__(256) vq_l2norm16_kernel(const void *x, int64_t N, int D, int64_t ldx, void *out, int64_t ldo)
{
    const int l16 = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool row_ok = row < N;
    const int64_t rowc = row_ok ? row : (N - 1);
    float xr[NK][4];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int d = 64 * k + 4 * l16;
#pragma unroll
        for (int e = 0; e < 4; ++e) xr[k][e] = 0.f;
        if (d < D) {
            if (XBF16) {
                const uint2 w = *(const uint2 *)((const unsigned short *)x + rowc * ldx + d);
                xr[k][0] = __uint_as_float(w.x << 16); xr[k][1] = __uint_as_float(w.x & 0xffff0000u);
                xr[k][2] = __uint_as_float(w.y << 16); xr[k][3] = __uint_as_float(w.y & 0xffff0000u);
            } else {
                const f32x4 w = *(const f32x4 *)((const float *)x + rowc * ldx + d);
                xr[k][0] = w.x; xr[k][1] = w.y; xr[k][2] = w.z; xr[k][3] = w.w;
            }
        }
    }
    const int T = D / 32;
    float ch[4] = {0.f, 0.f, 0.f, 0.f};                     // lanes 0..7: chain (l16 >> 1, l16 & 1, e)
#pragma unroll
    for (int t = 0; t < 2 * NK; ++t) {
        if (t < T) {
            const int src = (l16 & 7) + 8 * (t & 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = __shfl(xr[t >> 1][e], src, 16);
                ch[e] = __fadd_rn(ch[e], __fmul_rn(v, v));
            }
        }
    }
    const int hi = l16 & 1;
    float p[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float c0 = __shfl(ch[e], hi, 16), c1 = __shfl(ch[e], 2 + hi, 16), c2 = __shfl(ch[e], 4 + hi, 16), c3 = __shfl(ch[e], 6 + hi, 16);
        p[e] = __fadd_rn(__fadd_rn(__fadd_rn(c0, c1), c2), c3);
    }
    const float f_lo_own = __fadd_rn(__fadd_rn(__fadd_rn(p[0], p[1]), p[2]), p[3]);     // meaningful on even lanes (hi == 0)
    const float f_lo = __shfl(f_lo_own, 0, 16);
    const float f_hi_own = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(f_lo, p[0]), p[1]), p[2]), p[3]);   // meaningful on odd lanes
    const float x2 = __shfl(f_hi_own, 1, 16);
    float nrm = sqrtf(x2);
    if (XBF16) nrm = round_to_bf16(nrm);
    nrm = fmaxf(nrm, XBF16 ? round_to_bf16(1e-6f) : 1e-6f);
    if (!row_ok) return;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int d = 64 * k + 4 * l16;
        if (d >= D) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = xr[k][e] / nrm;
            if (XBF16) v[e] = round_to_bf16(v[e]);
        }
        if (XBF16) {
            uint2 w;
            w.x = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xffff0000u);
            w.y = (__float_as_uint(v[2]) >> 16) | (__float_as_uint(v[3]) & 0xffff0000u);
            *(uint2 *)((unsigned short *)out + row * ldo + d) = w;
        } else {
            *(f32x4 *)((float *)out + row * ldo + d) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

extern "C" int vqhip_l2norm_rows(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, void *out, int64_t ldo, void *stream)
{
    if (N < 0) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows: N < 0");
    if (N == 0) return 0;
    if (!x || !out) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows: null pointer");
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows: unknown dtype %d", x_dtype);
    if (vq_is_wide(D)) {
        if (ldx < D || ldo < D) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows: row stride smaller than D");
        return vq_wide_l2norm_rows(x, x_dtype, N, D, ldx, out, ldo, stream);
    }
    if (D != 32 && D != 64 && D != 128 && D != 256 && D != 512) VQ_FAIL(VQHIP_EDIM, "l2norm_rows: D=%d unsupported (32, 64, 128, 256, 512, or 513..2048)", D);
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    if (ldx < D || ldo < D) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows: row stride smaller than D");
    if ((((uintptr_t)x) % (4 * es)) || ((ldx * es) % (4 * es)) || (((uintptr_t)out) % (4 * es)) || ((ldo * es) % (4 * es)))
        VQ_FAIL(VQHIP_EALIGN, "l2norm_rows: rows must be aligned to 4 elements");
    hipStream_t st = (hipStream_t)stream;
    {
        static int old_kernel = -1;     // VQHIP_L2NORM_OLD=1: the 32-rows-per-wave kernel (A/B runs)
        if (old_kernel < 0) { const char *e = getenv("VQHIP_L2NORM_OLD"); old_kernel = (e && e[0] == '1') ? 1 : 0; }
        if (!old_kernel) {
            const unsigned b16 = (unsigned)((N + 15) / 16);
#define VQ_L2N16(NKV)                                                                                                              \
    do {                                                                                                                           \
        if (x_dtype == VQHIP_BF16) hipLaunchKernelGGL((vq_l2norm16_kernel<NKV, true>), dim3(b16), dim3(256), 0, st, x, N, D, ldx, out, ldo);  \
        else hipLaunchKernelGGL((vq_l2norm16_kernel<NKV, false>), dim3(b16), dim3(256), 0, st, x, N, D, ldx, out, ldo);            \
    } while (0)
            if (D <= 64) VQ_L2N16(1);
            else if (D == 128) VQ_L2N16(2);
            else if (D == 256) VQ_L2N16(4);
            else VQ_L2N16(8);
#undef VQ_L2N16
            return launch_status("vq_l2norm16_kernel");
        }
    }
    const unsigned blocks = (unsigned)((N + 127) / 128);
#define VQ_L2N(DTV)                                                                                                   \
    do {                                                                                                              \
        if (x_dtype == VQHIP_BF16) hipLaunchKernelGGL((vq_l2norm_kernel<DTV, true>), dim3(blocks), dim3(256), 0, st, x, N, ldx, out, ldo);  \
        else hipLaunchKernelGGL((vq_l2norm_kernel<DTV, false>), dim3(blocks), dim3(256), 0, st, x, N, ldx, out, ldo); \
    } while (0)
    switch (D) {
        case 32: VQ_L2N(32); break;
        case 64: VQ_L2N(64); break;
        case 128: VQ_L2N(128); break;
        case 256: VQ_L2N(256); break;
        default: VQ_L2N(512); break;
    }
#undef VQ_L2N
    return launch_status("vq_l2norm_kernel");
}

// backward of the row-wise l2norm, the gradient autograd derives for F.normalize (vqp.py:37-38: x / clamp_min(||x||, eps)):
//     gx = g / n  -  [||x|| >= eps] * x / ||x|| * sum_d(g_d x_d) / n^2,        n = max(||x||, eps)
// in ONE pass over x and g (autograd: the quotient's, the clamp's and the norm's backward, about ten elementwise kernels and
// reductions over N x D tensors).  16 lanes per row, 16 rows per workgroup of 256; any D <= 512 with D % 4 == 0; fp32 arithmetic,
// bf16 tensors: the norm rounded to bf16 as in the forward, the result rounded once.
template <int NK, bool XBF16>
__global__ void __launch_bounds__(256) vq_l2norm_bwd_kernel(const void *x, const void *g, int64_t N, int D, int64_t ldx, int64_t ldg,
                                                            void *out, int64_t ldo)
{
    const int l16 = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool row_ok = row < N;
    const int64_t rowc = row_ok ? row : (N - 1);
    float xr[NK][4], gr[NK][4];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int d = 64 * k + 4 * l16;
        const bool ok = d < D;
#pragma unroll
        for (int r = 0; r < 4; ++r) { xr[k][r] = 0.f; gr[k][r] = 0.f; }
        if (ok) {
            if (XBF16) {
                const uint2 wx = *(const uint2 *)((const unsigned short *)x + rowc * ldx + d);
                const uint2 wg = *(const uint2 *)((const unsigned short *)g + rowc * ldg + d);
                xr[k][0] = __uint_as_float(wx.x << 16); xr[k][1] = __uint_as_float(wx.x & 0xffff0000u);
                xr[k][2] = __uint_as_float(wx.y << 16); xr[k][3] = __uint_as_float(wx.y & 0xffff0000u);
                gr[k][0] = __uint_as_float(wg.x << 16); gr[k][1] = __uint_as_float(wg.x & 0xffff0000u);
                gr[k][2] = __uint_as_float(wg.y << 16); gr[k][3] = __uint_as_float(wg.y & 0xffff0000u);
            } else {
                const f32x4 wx = *(const f32x4 *)((const float *)x + rowc * ldx + d);
                const f32x4 wg = *(const f32x4 *)((const float *)g + rowc * ldg + d);
                xr[k][0] = wx.x; xr[k][1] = wx.y; xr[k][2] = wx.z; xr[k][3] = wx.w;
                gr[k][0] = wg.x; gr[k][1] = wg.y; gr[k][2] = wg.z; gr[k][3] = wg.w;
            }
        }
    }
    float x2 = 0.f, gx = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x2 = fmaf(xr[k][r], xr[k][r], x2);
            gx = fmaf(gr[k][r], xr[k][r], gx);
        }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        x2 += __shfl_xor(x2, o, 64);
        gx += __shfl_xor(gx, o, 64);
    }
    float nrm = sqrtf(x2);
    if (XBF16) nrm = round_to_bf16(nrm);
    const float eps = XBF16 ? round_to_bf16(1e-6f) : 1e-6f;
    const float n = fmaxf(nrm, eps);
    const float inv_n = 1.f / n;
    const float coef = (nrm >= eps && nrm > 0.f) ? (gx * inv_n * inv_n) / nrm : 0.f;
    if (!row_ok) return;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int d = 64 * k + 4 * l16;
        if (d >= D) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(-xr[k][r], coef, gr[k][r] * inv_n);
        if (XBF16) {
            uint2 w;
            w.x = (__float_as_uint(round_to_bf16(v[0])) >> 16) | (__float_as_uint(round_to_bf16(v[1])) & 0xffff0000u);
            w.y = (__float_as_uint(round_to_bf16(v[2])) >> 16) | (__float_as_uint(round_to_bf16(v[3])) & 0xffff0000u);
            *(uint2 *)((unsigned short *)out + row * ldo + d) = w;
        } else {
            *(f32x4 *)((float *)out + row * ldo + d) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

extern "C" int vqhip_l2norm_rows_bwd(const void *x, const void *g, int x_dtype, int64_t N, int D, int64_t ldx, int64_t ldg,
                                     void *out, int64_t ldo, void *stream)
{
    if (N < 0) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows_bwd: N < 0");
    if (N == 0) return 0;
    if (!x || !g || !out) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows_bwd: null pointer");
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows_bwd: unknown dtype %d", x_dtype);
    if (D < 4 || D > 512 || (D & 3)) VQ_FAIL(VQHIP_EDIM, "l2norm_rows_bwd: D=%d unsupported (multiples of 4 up to 512)", D);
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    const uintptr_t al = (uintptr_t)(4 * es) - 1;
    if ((((uintptr_t)x) & al) || (((uintptr_t)g) & al) || (((uintptr_t)out) & al) || ((ldx * es) & al) || ((ldg * es) & al) || ((ldo * es) & al))
        VQ_FAIL(VQHIP_EALIGN, "l2norm_rows_bwd: rows must be aligned to 4 elements");
    if (ldx < D || ldg < D || ldo < D) VQ_FAIL(VQHIP_EINVAL, "l2norm_rows_bwd: row stride smaller than D");
    const unsigned blocks = (unsigned)((N + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
#define VQ_L2B(NKV)                                                                                                              \
    do {                                                                                                                         \
        if (x_dtype == VQHIP_BF16) hipLaunchKernelGGL((vq_l2norm_bwd_kernel<NKV, true>), dim3(blocks), dim3(256), 0, st, x, g, N, D, ldx, ldg, out, ldo); \
        else hipLaunchKernelGGL((vq_l2norm_bwd_kernel<NKV, false>), dim3(blocks), dim3(256), 0, st, x, g, N, D, ldx, ldg, out, ldo); \
    } while (0)
    if (D <= 64) VQ_L2B(1);
    else if (D <= 128) VQ_L2B(2);
    else if (D <= 256) VQ_L2B(4);
    else VQ_L2B(8);
#undef VQ_L2B
    return launch_status("vq_l2norm_bwd_kernel");
}

// ------------------------------------------------------------------------------------------------
// exact pass over a LIST of rows (the rows vq_screen.hip could not certify).  Same arithmetic as
// vq_assign_kernel<DT, bf16, euclid> -- same device functions -- but organised for a short list: the codebook sweep
// is split over several workgroups per 128-row chunk (a few thousand rows would otherwise occupy a fraction of the
// CUs for a full sweep each), the partial winners meet in an atomicMin on the 64-bit key (bits(d) << 32 | index),
// which is exactly "smallest distance, then lowest index", and vq_finish_listed_kernel emits index, q and the
// squared error.  The list length is only known on the device: fixed grids, chunk loops.
// ------------------------------------------------------------------------------------------------
#ifndef VQ_REFINE_GRID
#define VQ_REFINE_GRID 1024
#endif
struct RefineArgs {
    const void *x;
    int64_t ldx;
    const float *packed;
    int C;
    int n_tiles;
    const int *row_list;
    const int *row_count;
    unsigned long long *keys;   // [list capacity], preset to ~0
    VqHeadStrides hs;           // batched heads: blockIdx.y = head
    // DIRECT (vq_tail_kernel): the sweep writes the index itself -- idx_out[row * idx_stride]; a split sweep counts its arrivals per
    // 128-row chunk in done[] (zeroed by the caller) and the last one reads the chunk's keys back
    int64_t *idx_out;
    int64_t idx_stride;
    int *done;
};

// (bid, nblk): this workgroup's number and the number of workgroups doing this work -- the kernel's own grid, or its share of
// the merged launch vq_tail_kernel further down
template <int DT, bool XBF16, int METRIC, bool DIRECT = false>
__device__ __forceinline__ void vq_refine_body(const RefineArgs &a0, char *smem, const unsigned bid, const unsigned nblk)
{
    RefineArgs a = a0;
    __shared__ int s_last;
    constexpr int TILE_F = 32 * DT + 256;
    constexpr int TILE_B = TILE_F * 4;
    constexpr int NCHUNK = TILE_B / 1024;
    constexpr int NG = DT / 8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int hi = lane >> 5;
    if (a0.hs.heads > 1) {
        const int64_t h = blockIdx.y;
        a.x = (const char *)a0.x + h * a0.hs.x;
        a.packed = (const float *)((const char *)a0.packed + h * a0.hs.packed);
        a.row_list = (const int *)((const char *)a0.row_list + h * a0.hs.ws);
        a.row_count = (const int *)((const char *)a0.row_count + h * a0.hs.ws);
        a.keys = (unsigned long long *)((char *)a0.keys + h * a0.hs.ws);
        if (DIRECT) {
            a.idx_out = (int64_t *)((char *)a0.idx_out + h * a0.hs.idx);
            a.done = (int *)((char *)a0.done + h * a0.hs.ws);
        }
    }
    const int list_n = __builtin_amdgcn_readfirstlane(*a.row_count);
    if (list_n <= 0) return;
    // the list length only exists on the device: a fixed 1-D grid, and the codebook sweep of every 128-row chunk is split over
    // as many workgroups as the grid has to spare (a short list would otherwise keep a handful of workgroups busy for a whole
    // sweep each: 70 us for a few hundred rows); the partial winners meet in the atomicMin below
    const int n_chunks = (list_n + VQHIP_ASSIGN_ROWS_PER_BLOCK - 1) / VQHIP_ASSIGN_ROWS_PER_BLOCK;
    int splits = (int)nblk / n_chunks;
    splits = splits < 1 ? 1 : (splits > a.n_tiles ? a.n_tiles : splits);
    const int tps = (a.n_tiles + splits - 1) / splits;   // tiles per split
    splits = (a.n_tiles + tps - 1) / tps;
    const int my_pieces = (NCHUNK - wave + 3) / 4;
    const int piece_off = wave * 1024 + lane * 16;

    for (int64_t w = bid; w < (int64_t)n_chunks * splits; w += nblk) {
        const int64_t chunk = w / splits;
        const int ct0 = (int)(w % splits) * tps;
        const int ct1 = min(a.n_tiles, ct0 + tps);
        __syncthreads();   // the previous chunk's last tile has been consumed by every wave
        for (int k = 0; k < my_pieces; ++k)
            *(f32x4 *)(smem + piece_off + k * 4096) =
                *(const f32x4 *)((const char *)a.packed + (size_t)ct0 * TILE_B + piece_off + (size_t)k * 4096);

        const int64_t pos = chunk * VQHIP_ASSIGN_ROWS_PER_BLOCK + wave * 32 + j;
        const bool row_ok = pos < list_n;
        const int64_t row = a.row_list[row_ok ? pos : (int64_t)(list_n - 1)];
        float xr[DT / 2];   // load layout, as in vq_assign_kernel
        if (XBF16) {
            const uint2 *p = (const uint2 *)((const unsigned short *)a.x + row * a.ldx + 4 * hi);
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                const uint2 w = p[m * 2];
                xr[4 * m + 0] = __uint_as_float(w.x << 16);
                xr[4 * m + 1] = __uint_as_float(w.x & 0xffff0000u);
                xr[4 * m + 2] = __uint_as_float(w.y << 16);
                xr[4 * m + 3] = __uint_as_float(w.y & 0xffff0000u);
            }
        } else {
            const f32x4 *p = (const f32x4 *)((const float *)a.x + row * a.ldx + 4 * hi);
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                const f32x4 w = p[m * 2];
                xr[4 * m + 0] = w.x;
                xr[4 * m + 1] = w.y;
                xr[4 * m + 2] = w.z;
                xr[4 * m + 3] = w.w;
            }
        }
        const float x2 = (METRIC == 0) ? x2_aten_order<DT>(xr, j) : 0.f;   // cosine (rows already unit-norm): raw dot products
#pragma unroll
        for (int m = 0; m < NG; ++m) {   // -> MFMA B-operand layout
            swap32(xr[4 * m + 0], xr[4 * m + 1]);
            swap32(xr[4 * m + 2], xr[4 * m + 3]);
        }

        float bd = (METRIC == 0) ? INFINITY : -INFINITY, bs = INFINITY;
        int bi = 0;
        for (int ct = ct0; ct < ct1; ++ct) {
            const int buf = (ct - ct0) & 1;
            __syncthreads();
            const char *tile = smem + buf * TILE_B;
            const f32x4 *ap = (const f32x4 *)tile + (hi * 32 + j);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const bool more = ct + 1 < ct1;
            mfma_sweep_tile<DT>(ap, xr, acc, (const char *)a.packed + (size_t)(more ? ct + 1 : ct) * TILE_B + piece_off,
                                smem + (buf ^ 1) * TILE_B + piece_off, more ? my_pieces : 0);
            argmin_tile<METRIC>(acc, (const float *)tile + 32 * DT, x2, ct * 32, hi, a.C, bd, bs, bi);
        }
        {
            const float od = __shfl_xor(bd, 32, 64);
            const int oi = __shfl_xor(bi, 32, 64);
            const bool take = (METRIC == 0) ? ((od < bd) || (od == bd && oi < bi)) : ((od > bd) || (od == bd && oi < bi));
            bd = take ? od : bd;
            bi = take ? oi : bi;
        }
        // key order = (better score, then lower index): distances are >= 0, so their bit patterns order like the values;
        // similarities go through the usual sign fix (monotone map to uint) and are inverted so that larger is smaller
        unsigned hk = __float_as_uint(bd);
        if (METRIC != 0) hk = ~(hk ^ ((hk >> 31) ? 0xffffffffu : 0x80000000u));
        if (DIRECT && splits == 1) {                      // this workgroup swept the whole codebook for the chunk: the winner is final
            if (row_ok && hi == 0) a.idx_out[row * a.idx_stride] = (int64_t)bi;
            continue;
        }
        if (row_ok && hi == 0)
            atomicMin(a.keys + pos, ((unsigned long long)hk << 32) | (unsigned long long)(unsigned)bi);
        if (DIRECT) {
            // the chunk's last arriver (of `splits` workgroups) reads the merged keys back (device-scope atomics: L2) and writes the indices
            __threadfence();
            __syncthreads();
            if (tid == 0) s_last = (atomicAdd(a.done + chunk, 1) == splits - 1) ? 1 : 0;
            __syncthreads();
            if (s_last) {
                __threadfence();
                const int64_t p2 = chunk * VQHIP_ASSIGN_ROWS_PER_BLOCK + tid;
                if (tid < VQHIP_ASSIGN_ROWS_PER_BLOCK && p2 < list_n) {
                    const unsigned long long k = __hip_atomic_load(a.keys + p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a.idx_out[(int64_t)a.row_list[p2] * a.idx_stride] = (int64_t)(unsigned)(k & 0xffffffffull);
                }
            }
        }
    }
}

template <int DT, bool XBF16, int METRIC>
__global__ void __launch_bounds__(256, (DT <= 256 ? 2 : 1)) vq_refine_kernel(const RefineArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    vq_refine_body<DT, XBF16, METRIC>(a, smem, blockIdx.x, gridDim.x);
}

struct FinishArgs {
    const void *x;
    int64_t ldx;
    const void *codes;       // bf16 rows: the bf16 codebook copy; fp32 rows: embed
    int D;
    const int *row_list;
    const int *row_count;    // [0] rows of the full exact pass (list front), [1] rows of the pair pass (list back, from cap - 1 down)
    int64_t cap;             // list capacity (N)
    const unsigned long long *keys;
    int64_t *idx_out;
    int64_t idx_stride;      // idx_out[row * idx_stride]
    void *q_out;             // nullable, x's dtype
    int64_t ldq;
    void *resid_out;         // nullable, x's dtype: x - q
    int64_t ldr;
    double *sqerr_partial;   // nullable, one entry per workgroup
    const uint8_t *row_mask;
    VqHeadStrides hs;        // batched heads: blockIdx.y = head (no residual / squared-error outputs then)
};

// one wave per listed row: idx, q row, sum (q - x)^2.  A row is a chain of dependent loads (list -> key -> code row): the kernel is
// latency bound, so VQ_FINISH_WAVES waves per workgroup (8192 waves in the grid) keep the rows per wave at a handful
#define VQ_FINISH_WAVES 16
template <bool XBF16>
__global__ void __launch_bounds__(VQ_FINISH_WAVES * 64) vq_finish_listed_kernel(const FinishArgs a0)
{
    __shared__ double red[VQ_FINISH_WAVES];
    FinishArgs a = a0;
    if (a0.hs.heads > 1) {
        const int64_t h = blockIdx.y;
        a.x = (const char *)a0.x + h * a0.hs.x;
        a.codes = (const char *)a0.codes + h * a0.hs.codes;
        a.row_list = (const int *)((const char *)a0.row_list + h * a0.hs.ws);
        a.row_count = (const int *)((const char *)a0.row_count + h * a0.hs.ws);
        a.keys = (const unsigned long long *)((const char *)a0.keys + h * a0.hs.ws);
        a.idx_out = (int64_t *)((char *)a0.idx_out + h * a0.hs.idx);
        if (a0.q_out) a.q_out = (char *)a0.q_out + h * a0.hs.q;
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_full = a.row_count[0];
    const int n = n_full + a.row_count[1];
    double acc = 0.0;
    for (int64_t v = (int64_t)blockIdx.x * VQ_FINISH_WAVES + wave; v < n; v += (int64_t)gridDim.x * VQ_FINISH_WAVES) {
        const int64_t pos = v < n_full ? v : a.cap - 1 - (v - n_full);
        const int64_t row = a.row_list[pos];
        const int idx = (int)(unsigned)(a.keys[pos] & 0xffffffffull);
        if (lane == 0) a.idx_out[row * a.idx_stride] = (int64_t)idx;
        float ls = 0.f;
        for (int c0 = lane * 4; c0 < a.D; c0 += 256) {
            float d0, d1, d2, d3;
            if (XBF16) {
                const uint2 g = *(const uint2 *)((const unsigned short *)a.codes + (size_t)idx * a.D + c0);
                const uint2 xv = *(const uint2 *)((const unsigned short *)a.x + row * a.ldx + c0);
                if (a.q_out) *(uint2 *)((unsigned short *)a.q_out + row * a.ldq + c0) = g;
                if (a.resid_out) *(uint2 *)((unsigned short *)a.resid_out + row * a.ldr + c0) = vq_bf16x4_sub(xv, g);
                d0 = __uint_as_float(g.x << 16) - __uint_as_float(xv.x << 16);
                d1 = __uint_as_float(g.x & 0xffff0000u) - __uint_as_float(xv.x & 0xffff0000u);
                d2 = __uint_as_float(g.y << 16) - __uint_as_float(xv.y << 16);
                d3 = __uint_as_float(g.y & 0xffff0000u) - __uint_as_float(xv.y & 0xffff0000u);
            } else {
                const f32x4 g = *(const f32x4 *)((const float *)a.codes + (size_t)idx * a.D + c0);
                const f32x4 xv = *(const f32x4 *)((const float *)a.x + row * a.ldx + c0);
                if (a.q_out) *(f32x4 *)((float *)a.q_out + row * a.ldq + c0) = g;
                if (a.resid_out) *(f32x4 *)((float *)a.resid_out + row * a.ldr + c0) = xv - g;
                d0 = g.x - xv.x; d1 = g.y - xv.y; d2 = g.z - xv.z; d3 = g.w - xv.w;
            }
            ls += ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
        }
        double ds = (double)ls;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ds += __shfl_xor(ds, o, 64);
        if (!a.row_mask || a.row_mask[row] != 0) acc += ds;
    }
    if (a.sqerr_partial) {
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < VQ_FINISH_WAVES; ++w) t += red[w];
            a.sqerr_partial[blockIdx.x] = t;
        }
    }
}

template <int DT, bool XBF16, int METRIC>
static int launch_refine(const RefineArgs &a, unsigned gx, hipStream_t st)
{
    constexpr int SMEM = 2 * (32 * DT + 256) * 4;
    static VqAttrOnce once;
    if (int rc = vq_set_max_smem(once, (const void *)vq_refine_kernel<DT, XBF16, METRIC>, SMEM, "vq_refine_kernel")) return rc;
    hipLaunchKernelGGL((vq_refine_kernel<DT, XBF16, METRIC>), dim3(gx, a.hs.heads > 1 ? a.hs.heads : 1), dim3(256), SMEM, st, a);
    return launch_status("vq_refine_kernel");
}

template <int DT>
static int dispatch_refine(const RefineArgs &a, int x_dtype, int metric, unsigned gx, hipStream_t st)
{
    if (metric == VQHIP_EUCLID)
        return x_dtype == VQHIP_BF16 ? launch_refine<DT, true, 0>(a, gx, st) : launch_refine<DT, false, 0>(a, gx, st);
    return x_dtype == VQHIP_BF16 ? launch_refine<DT, true, 1>(a, gx, st) : launch_refine<DT, false, 1>(a, gx, st);
}

// Rows whose winner is one of TWO known codes (the screen's best and runner-up, every third code certified out):
// both distances in the reference's own arithmetic -- ATen-order ||x||^2, ||c||^2 from the packed codebook, x.c as ONE fp32
// FMA chain in ascending k (what the exact kernels' f32 MFMAs and oracle/vq_oracle.c compute), (x2 + y2) + (-2 xy),
// clamp, correctly rounded sqrt -- smaller distance wins, the lower index on a tie (vqp.py:58-62, 140).  Cosine: the two
// dot products of the unit-norm row, larger wins.  One row per lane; the winner replaces the candidates in keys[pos].
struct PairArgs {
    const void *x;
    int64_t ldx;
    const float *embed;
    const float *packed;      // exact section: ||c||^2 sits behind each 32-code tile
    int D;
    const int *row_list;
    const int *row_count;     // [1] = number of pair rows, stored at list positions cap - 1 - p
    int64_t cap;
    unsigned long long *keys;
    VqHeadStrides hs;         // batched heads: blockIdx.y = head
    int64_t *idx_out;         // DIRECT (vq_tail_kernel): the winner goes straight to idx_out[row * idx_stride]
    int64_t idx_stride;
};

// The 16-byte pieces a lane streams from its OWN row and its two code rows would touch 64 different cache lines per wave
// instruction and thrash the 16 KiB vector L1 (every piece re-fetched from L2: the first version ran at 30-38 us for ~30k rows).
// So the wave loads 32-element pieces of its 64 rows COOPERATIVELY (8 lanes per fp32 piece: whole 128-byte lines), parks them
// in LDS (row pitch + 16 bytes: conflict-free b128 reads), and every lane then reads its own row's piece back.  The next
// piece's global loads are in flight while the current one is multiplied.  Arithmetic unchanged: one ascending FMA chain per code.
#define VQ_PAIR_WAVES 2
template <bool XBF16> struct PairCfg {
    static constexpr int KC = 32;
    static constexpr int EP = KC * 4 + 16;                          // LDS row pitch of a code piece (bytes)
    static constexpr int XP = (XBF16 ? KC * 2 : KC * 4) + 16;       // ... of a row piece
    static constexpr int WAVE_B = 64 * (2 * EP + XP);
    static constexpr int SMEM = VQ_PAIR_WAVES * WAVE_B;
};

template <int DT, bool XBF16, int METRIC, bool DIRECT = false>
__device__ __forceinline__ void vq_pair_body(const PairArgs &a0, char *smem, const unsigned bid, const unsigned nblk)
{
    PairArgs a = a0;
    if (a0.hs.heads > 1) {
        const int64_t h = blockIdx.y;
        if (DIRECT) a.idx_out = (int64_t *)((char *)a0.idx_out + h * a0.hs.idx);
        a.x = (const char *)a0.x + h * a0.hs.x;
        a.embed = (const float *)((const char *)a0.embed + h * a0.hs.embed);
        a.packed = (const float *)((const char *)a0.packed + h * a0.hs.packed);
        a.row_list = (const int *)((const char *)a0.row_list + h * a0.hs.ws);
        a.row_count = (const int *)((const char *)a0.row_count + h * a0.hs.ws);
        a.keys = (unsigned long long *)((char *)a0.keys + h * a0.hs.ws);
    }
    constexpr int KC = PairCfg<XBF16>::KC, NCH = DT / KC;
    constexpr int EP = PairCfg<XBF16>::EP;
    constexpr int XP = PairCfg<XBF16>::XP;
    constexpr int NXL = XBF16 ? 4 : 8;                       // wave loads per row piece of the 64 rows
    constexpr int XLPR = 64 / (64 / NXL) ;                   // lanes per row piece: 4 (bf16: 64 B) or 8 (fp32: 128 B)
    constexpr int WAVE_B = PairCfg<XBF16>::WAVE_B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= VQ_PAIR_WAVES) return;
    char *se1 = smem + wave * WAVE_B, *se2 = se1 + 64 * EP, *sx = se2 + 64 * EP;
    const int n = a.row_count[1];
    for (int64_t base = ((int64_t)bid * VQ_PAIR_WAVES + wave) * 64; base < n; base += (int64_t)nblk * VQ_PAIR_WAVES * 64) {
        const int64_t p = (base + lane < n) ? base + lane : (int64_t)n - 1;
        const int64_t pos = a.cap - 1 - p;
        const int64_t row = a.row_list[pos];
        const unsigned long long cand = a.keys[pos];
        const int c1 = (int)(unsigned)(cand & 0xffffffffull), c2 = (int)(unsigned)(cand >> 32);
        // cooperative pieces: wave load i of a code stream covers the rows 8 i .. 8 i + 7 (8 lanes x 16 bytes each)
        unsigned oe1[8], oe2[8];          // byte offsets into embed (C * D * 4 < 2^32): uniform base + 32-bit lane offset
        const char *px[NXL];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = i * 8 + (lane >> 3);
            oe1[i] = (unsigned)__shfl(c1, r, 64) * (unsigned)(DT * 4) + (unsigned)(lane & 7) * 16u;
            oe2[i] = (unsigned)__shfl(c2, r, 64) * (unsigned)(DT * 4) + (unsigned)(lane & 7) * 16u;
        }
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            const int r = i * (64 / XLPR) + lane / XLPR;
            const int64_t rr = ((int64_t)__shfl((int)(row >> 32), r, 64) << 32) | (int64_t)(unsigned)__shfl((int)(row & 0xffffffffll), r, 64);
            px[i] = (const char *)a.x + (rr * a.ldx) * (XBF16 ? 2 : 4) + (lane % XLPR) * 16;
        }
        f32x4 g1[8], g2[8], gx[NXL];
        auto issue = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                g1[i] = *(const f32x4 *)((const char *)a.embed + (size_t)(oe1[i] + (unsigned)k0 * 4u));
                g2[i] = *(const f32x4 *)((const char *)a.embed + (size_t)(oe2[i] + (unsigned)k0 * 4u));
            }
#pragma unroll
            for (int i = 0; i < NXL; ++i) gx[i] = *(const f32x4 *)(px[i] + (size_t)k0 * (XBF16 ? 2 : 4));
        };
        issue(0);
        float xy1 = 0.f, xy2 = 0.f;
        float ch[32];                      // ATen-order ||x||^2: 32 interleaved chains, combined below (aten_sumsq_seq's order)
#pragma unroll
        for (int c = 0; c < 32; ++c) ch[c] = 0.f;
#pragma unroll 1
        for (int k = 0; k < NCH; ++k) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                *(f32x4 *)(se1 + (i * 8 + (lane >> 3)) * EP + (lane & 7) * 16) = g1[i];
                *(f32x4 *)(se2 + (i * 8 + (lane >> 3)) * EP + (lane & 7) * 16) = g2[i];
            }
#pragma unroll
            for (int i = 0; i < NXL; ++i) *(f32x4 *)(sx + (i * (64 / XLPR) + lane / XLPR) * XP + (lane % XLPR) * 16) = gx[i];
            if (k + 1 < NCH) issue((k + 1) * KC);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // this lane's row piece against its two code pieces, 4 elements at a time in ascending k (ONE FMA chain per code)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float w0, w1, w2, w3;
                if (XBF16) {
                    const uint2 w = *(const uint2 *)(sx + lane * XP + q * 8);
                    w0 = __uint_as_float(w.x << 16); w1 = __uint_as_float(w.x & 0xffff0000u);
                    w2 = __uint_as_float(w.y << 16); w3 = __uint_as_float(w.y & 0xffff0000u);
                } else {
                    const f32x4 w = *(const f32x4 *)(sx + lane * XP + q * 16);
                    w0 = w.x; w1 = w.y; w2 = w.z; w3 = w.w;
                }
                const f32x4 u1 = *(const f32x4 *)(se1 + lane * EP + q * 16), u2 = *(const f32x4 *)(se2 + lane * EP + q * 16);
                xy1 = __builtin_fmaf(w0, u1.x, xy1); xy1 = __builtin_fmaf(w1, u1.y, xy1);
                xy1 = __builtin_fmaf(w2, u1.z, xy1); xy1 = __builtin_fmaf(w3, u1.w, xy1);
                xy2 = __builtin_fmaf(w0, u2.x, xy2); xy2 = __builtin_fmaf(w1, u2.y, xy2);
                xy2 = __builtin_fmaf(w2, u2.z, xy2); xy2 = __builtin_fmaf(w3, u2.w, xy2);
                if (METRIC == 0) {
                    ch[4 * q + 0] += w0 * w0; ch[4 * q + 1] += w1 * w1; ch[4 * q + 2] += w2 * w2; ch[4 * q + 3] += w3 * w3;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // this piece has been read before the next one is parked
            __builtin_amdgcn_wave_barrier();
        }
        int win;
        if (METRIC == 0) {
            float x2 = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) x2 += ((ch[l] + ch[8 + l]) + ch[16 + l]) + ch[24 + l];
            const int tf = 32 * DT + 256;
            const float y1 = a.packed[(size_t)(c1 >> 5) * tf + 32 * DT + (c1 & 31)], y2 = a.packed[(size_t)(c2 >> 5) * tf + 32 * DT + (c2 & 31)];
            const float s1 = __builtin_fmaf(-2.f, xy1, x2 + y1), s2 = __builtin_fmaf(-2.f, xy2, x2 + y2);
            const float d1 = sqrtf(fmaxf(s1, 1e-8f)), d2 = sqrtf(fmaxf(s2, 1e-8f));
            win = (d2 < d1 || (d2 == d1 && c2 < c1)) ? c2 : c1;
        } else {
            win = (xy2 > xy1 || (xy2 == xy1 && c2 < c1)) ? c2 : c1;
        }
        if (base + lane < n) {
            if (DIRECT) a.idx_out[row * a.idx_stride] = (int64_t)win;
            else a.keys[pos] = (unsigned long long)(unsigned)win;
        }
    }
}

template <int DT, bool XBF16, int METRIC>
__global__ void __launch_bounds__(VQ_PAIR_WAVES * 64) vq_pair_kernel(const PairArgs a)
{
    __shared__ __attribute__((aligned(16))) char smem[PairCfg<XBF16>::SMEM];
    vq_pair_body<DT, XBF16, METRIC>(a, smem, blockIdx.x, gridDim.x);
}

template <int DT>
static int dispatch_pair(const PairArgs &a, int x_dtype, int metric, unsigned blocks, hipStream_t st)
{
#define VQ_PAIR(B, M) hipLaunchKernelGGL((vq_pair_kernel<DT, B, M>), dim3(blocks, a.hs.heads > 1 ? a.hs.heads : 1), dim3(VQ_PAIR_WAVES * 64), 0, st, a)
    if (metric == VQHIP_EUCLID) { if (x_dtype == VQHIP_BF16) VQ_PAIR(true, 0); else VQ_PAIR(false, 0); }
    else                        { if (x_dtype == VQHIP_BF16) VQ_PAIR(true, 1); else VQ_PAIR(false, 1); }
#undef VQ_PAIR
    return launch_status("vq_pair_kernel");
}

// ONE launch for the listed exact passes of a residual-chain stage (index output only): workgroups [0, gx) run the exact sweep of the
// open rows, the others the two exact distances of the pair rows, and both write their winners to idx_out themselves -- was three
// dependent launches (refine -> pair -> finish, ~50 us of a stage that the next stage's screening kernel waits for).
template <int DT, bool XBF16, int METRIC>
__global__ void __launch_bounds__(256, (DT <= 256 ? 2 : 1)) vq_tail_kernel(const RefineArgs r, const PairArgs p, const unsigned gx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x < gx) vq_refine_body<DT, XBF16, METRIC, true>(r, smem, blockIdx.x, gx);
    else vq_pair_body<DT, XBF16, METRIC, true>(p, smem, blockIdx.x - gx, gridDim.x - gx);
}

// Built, bit-identical (test_residual_chain_merged_exact_passes_...), and MEASURED SLOWER than the three launches it replaces: cfg 3
// 2.80 -> 2.96 ms, cfg 5 14.47 -> 14.75 ms (profiles/r6_ab/summary.md, r6g: one box, same run; round 3 had found refine + pair merged "a wash"): the
// pair rows' workgroups take the sweep's 256 registers and 34 - 68 KiB of LDS, i.e. two per CU, and every split sweep pays two barriers
// and a device-scope counter.  OFF unless VQHIP_TAIL=1.
int vq_tail_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("VQHIP_TAIL"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

template <int DT, bool XBF16, int METRIC>
static int launch_tail(const RefineArgs &r, const PairArgs &p, unsigned gx, unsigned pb, hipStream_t st)
{
    constexpr int SM_R = 2 * (32 * DT + 256) * 4, SM_P = PairCfg<XBF16>::SMEM, SMEM = SM_R > SM_P ? SM_R : SM_P;
    static VqAttrOnce once;
    if (int rc = vq_set_max_smem(once, (const void *)vq_tail_kernel<DT, XBF16, METRIC>, SMEM, "vq_tail_kernel")) return rc;
    hipLaunchKernelGGL((vq_tail_kernel<DT, XBF16, METRIC>), dim3(gx + pb, r.hs.heads > 1 ? r.hs.heads : 1), dim3(256), SMEM, st, r, p, gx);
    return launch_status("vq_tail_kernel");
}

template <int DT>
static int dispatch_tail(const RefineArgs &r, const PairArgs &p, int x_dtype, int metric, unsigned gx, unsigned pb, hipStream_t st)
{
    if (metric == VQHIP_EUCLID)
        return x_dtype == VQHIP_BF16 ? launch_tail<DT, true, 0>(r, p, gx, pb, st) : launch_tail<DT, false, 0>(r, p, gx, pb, st);
    return x_dtype == VQHIP_BF16 ? launch_tail<DT, true, 1>(r, p, gx, pb, st) : launch_tail<DT, false, 1>(r, p, gx, pb, st);
}

int vq_assign_listed_direct(const void *x, int x_dtype, int metric, int64_t N, int D, int64_t ldx, const float *packed, const float *embed, int C,
                            int64_t *idx_out, int64_t idx_stride, const int *row_list, const int *row_count, unsigned long long *keys, int *done,
                            hipStream_t st, const VqHeadStrides *hs)
{
    VqHeadStrides h1;
    h1.heads = 1; h1.x = h1.packed = h1.embed = h1.codes = h1.idx = h1.q = h1.ws = 0;
    const VqHeadStrides &H = hs ? *hs : h1;
    RefineArgs r;
    r.x = x; r.ldx = ldx; r.packed = packed; r.C = C; r.n_tiles = (C + 31) / 32;
    r.row_list = row_list; r.row_count = row_count; r.keys = keys; r.hs = H;
    r.idx_out = idx_out; r.idx_stride = idx_stride; r.done = done;
    const int64_t chunks = vqhip_assign_blocks(N);
    const int64_t want = chunks * r.n_tiles;
    const unsigned gx = (unsigned)(want < VQ_REFINE_GRID ? want : VQ_REFINE_GRID);
    PairArgs pa;
    pa.x = x; pa.ldx = ldx; pa.embed = embed; pa.packed = packed; pa.D = D;
    pa.row_list = row_list; pa.row_count = row_count; pa.cap = N; pa.keys = keys; pa.hs = H;
    pa.idx_out = idx_out; pa.idx_stride = idx_stride;
    const int64_t pb = (N + VQ_PAIR_WAVES * 64 - 1) / (VQ_PAIR_WAVES * 64);
    const unsigned blocks = (unsigned)(pb < 1024 ? pb : 1024);
    switch (pick_dt(D)) {
        case 32: return dispatch_tail<32>(r, pa, x_dtype, metric, gx, blocks, st);
        case 64: return dispatch_tail<64>(r, pa, x_dtype, metric, gx, blocks, st);
        case 128: return dispatch_tail<128>(r, pa, x_dtype, metric, gx, blocks, st);
        case 256: return dispatch_tail<256>(r, pa, x_dtype, metric, gx, blocks, st);
        case 512: return dispatch_tail<512>(r, pa, x_dtype, metric, gx, blocks, st);
        default: VQ_FAIL(VQHIP_EDIM, "assign_listed_direct: D=%d unsupported", D);
    }
}

int vq_assign_listed(const void *x, int x_dtype, int metric, int64_t N, int D, int64_t ldx, const float *packed, const float *embed, int C,
                     int64_t *idx_out, int64_t idx_stride, void *q_out, int64_t ldq, void *resid_out, int64_t ldr, double *sqerr_partial,
                     const uint8_t *row_mask, const int *row_list, const int *row_count, unsigned long long *keys, int with_pairs,
                     hipStream_t st, const VqHeadStrides *hs)
{
    VqHeadStrides h1;
    h1.heads = 1; h1.x = h1.packed = h1.embed = h1.codes = h1.idx = h1.q = h1.ws = 0;
    const VqHeadStrides &H = hs ? *hs : h1;
    if (H.heads > 1 && (resid_out || sqerr_partial)) VQ_FAIL(VQHIP_EINVAL, "assign_listed: no residual / squared-error outputs in a batched launch");
    // keys[0 .. row_count[0]) were preset to ~0 by whoever built the list (the screening kernels); the pair entries at
    // [N - row_count[1], N) hold their two candidates
    RefineArgs r;
    r.x = x; r.ldx = ldx; r.packed = packed; r.C = C; r.n_tiles = (C + 31) / 32;
    r.row_list = row_list; r.row_count = row_count; r.keys = keys; r.hs = H;
    r.idx_out = nullptr; r.idx_stride = 1; r.done = nullptr;
    const int64_t chunks = vqhip_assign_blocks(N);
    const int64_t want = chunks * r.n_tiles;              // one workgroup per (chunk, tile) at most
    const unsigned gx = (unsigned)(want < VQ_REFINE_GRID ? want : VQ_REFINE_GRID);
    int rc;
    // (Both exact passes side by side in ONE launch -- workgroups [0, gx) sweep, the rest decide pairs -- measured a wash in round 3:
    //  cfg 2 +0.9 %, cfg 3 -1.4 %, cfg 5 +1.8 %; the pair rows' workgroups inherit the sweep's LDS and register budget.  Removed.)
    PairArgs pa;
    pa.x = x; pa.ldx = ldx; pa.embed = embed; pa.packed = packed; pa.D = D;
    pa.row_list = row_list; pa.row_count = row_count; pa.cap = N; pa.keys = keys; pa.hs = H;
    pa.idx_out = nullptr; pa.idx_stride = 1;
    const int64_t pb = (N + VQ_PAIR_WAVES * 64 - 1) / (VQ_PAIR_WAVES * 64);
    const unsigned blocks = (unsigned)(pb < 1024 ? pb : 1024);
    switch (pick_dt(D)) {
        case 32: rc = dispatch_refine<32>(r, x_dtype, metric, gx, st); break;
        case 64: rc = dispatch_refine<64>(r, x_dtype, metric, gx, st); break;
        case 128: rc = dispatch_refine<128>(r, x_dtype, metric, gx, st); break;
        case 256: rc = dispatch_refine<256>(r, x_dtype, metric, gx, st); break;
        case 512: rc = dispatch_refine<512>(r, x_dtype, metric, gx, st); break;
        default: VQ_FAIL(VQHIP_EDIM, "assign_listed: D=%d unsupported", D);
    }
    if (rc) return rc;
    if (with_pairs) {
        switch (pick_dt(D)) {
            case 32: rc = dispatch_pair<32>(pa, x_dtype, metric, blocks, st); break;
            case 64: rc = dispatch_pair<64>(pa, x_dtype, metric, blocks, st); break;
            case 128: rc = dispatch_pair<128>(pa, x_dtype, metric, blocks, st); break;
            case 256: rc = dispatch_pair<256>(pa, x_dtype, metric, blocks, st); break;
            default: rc = dispatch_pair<512>(pa, x_dtype, metric, blocks, st); break;
        }
        if (rc) return rc;
    }
    const unsigned nh = (unsigned)(H.heads > 1 ? H.heads : 1);
    FinishArgs f;
    f.x = x; f.ldx = ldx;
    f.codes = (x_dtype == VQHIP_BF16) ? (const void *)((const char *)packed + packed_bf16_offset(C, D)) : (const void *)embed;
    f.D = D; f.row_list = row_list; f.row_count = row_count; f.cap = N; f.keys = keys;
    f.idx_out = idx_out; f.idx_stride = idx_stride; f.q_out = q_out; f.ldq = ldq; f.resid_out = resid_out; f.ldr = ldr; f.sqerr_partial = sqerr_partial; f.row_mask = row_mask; f.hs = H;
    if (x_dtype == VQHIP_BF16)
        hipLaunchKernelGGL(vq_finish_listed_kernel<true>, dim3(VQ_FINISH_BLOCKS, nh), dim3(VQ_FINISH_WAVES * 64), 0, st, f);
    else
        hipLaunchKernelGGL(vq_finish_listed_kernel<false>, dim3(VQ_FINISH_BLOCKS, nh), dim3(VQ_FINISH_WAVES * 64), 0, st, f);
    return launch_status("vq_finish_listed_kernel");
}

// ------------------------------------------------------------------------------------------------
// fused residual VQ (rvq.py:469-568): Q nearest-code sweeps on the running residual, the residual rows
// never leave the VGPRs between stages.  Euclidean metric, uniform codebook size.
//   per stage q:  idx[n, q] = argmin_c cdist(res, embed_q);  g = embed_q[idx]  (rounded to bf16 for bf16 I/O)
//                 loss_q += sum (g - res)^2 ;  res <- res - g   (rvq.py:524, exact fp32 / bf16 arithmetic of the
//                 reference: quantized and residual are x-dtype tensors there)
// Optional resid_out [N, Q, D] (x dtype) receives the INPUT of every stage: the EMA statistics
// (vqp.py:602-606 per layer) and the shared-codebook expiry (rvq.py:600-601) consume it afterwards.
// quantized_out = sum_q g_q is produced by vq_decode_kernel from the indices (same q-order running sum as
// rvq.py:525).  Masked rows (row_mask == 0): index -1, contribution 0, excluded from the loss.
// ------------------------------------------------------------------------------------------------
struct RvqArgs {
    const void *x;
    int64_t N;
    int D;
    int64_t ldx;
    const float *packed;
    int64_t packed_qstride;  // floats between the packed codebooks of consecutive stages (0: shared)
    const float *embed;
    int64_t embed_qstride;   // floats between codebooks (0: shared)
    size_t bf16_off;         // byte offset of the bf16 codebook copy inside each stage's packed buffer
    int C;
    int n_tiles;
    int Q;
    int64_t *idx_out;        // [N, Q]
    void *resid_out;         // nullable [N, Q, D] in x's dtype
    double *sqerr_partial;   // nullable [Q, gridDim.x]
    const uint8_t *row_mask;
    int x_vec;
};

// VEC: D == DT and every row is vector-aligned (an instantiation of its own: with the element-wise fallback in the same body the
// D = 256 / 512 kernels spilled 400 - 1 000 registers)
template <int DT, bool XBF16, bool VEC>
__global__ void __launch_bounds__(256, (DT <= 256 ? 2 : 1)) vq_rvq_kernel(const RvqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TILE_F = 32 * DT + 256;
    constexpr int TILE_B = TILE_F * 4;
    constexpr int NCHUNK = TILE_B / 1024;
    constexpr int NG = DT / 8;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int hi = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * VQHIP_ASSIGN_ROWS_PER_BLOCK + wave * 32 + j;
    const bool row_ok = row < a.N;
    const int64_t rowc = row_ok ? row : (a.N - 1);
    const bool live = row_ok && (!a.row_mask || a.row_mask[rowc] != 0);

    const int nt = a.n_tiles;
    const int total_tiles = nt * a.Q;
    const int my_pieces = (NCHUNK - wave + 3) / 4;
    const int piece_off = wave * 1024 + lane * 16;
    auto tile_src = [&](int gt) {   // gt = global tile counter over (stage, tile); LDS buffer = gt & 1
        const int q = gt / nt, ct = gt - q * nt;
        return (const char *)(a.packed + (size_t)q * a.packed_qstride) + (size_t)ct * TILE_B + piece_off;
    };
    for (int k = 0; k < my_pieces; ++k)
        *(f32x4 *)(smem + piece_off + k * 4096) = *(const f32x4 *)(tile_src(0) + (size_t)k * 4096);

    float xr[DT / 2];   // the running residual, load layout between stages, B-operand layout inside a sweep
    if (VEC) {
        if (XBF16) {
            const uint2 *p = (const uint2 *)((const unsigned short *)a.x + rowc * a.ldx + 4 * hi);
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                const uint2 w = p[m * 2];
                xr[4 * m + 0] = __uint_as_float(w.x << 16);
                xr[4 * m + 1] = __uint_as_float(w.x & 0xffff0000u);
                xr[4 * m + 2] = __uint_as_float(w.y << 16);
                xr[4 * m + 3] = __uint_as_float(w.y & 0xffff0000u);
            }
        } else {
            const f32x4 *p = (const f32x4 *)((const float *)a.x + rowc * a.ldx + 4 * hi);
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                const f32x4 w = p[m * 2];
                xr[4 * m + 0] = w.x; xr[4 * m + 1] = w.y; xr[4 * m + 2] = w.z; xr[4 * m + 3] = w.w;
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < NG; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 8 * m + 4 * hi + r;
                xr[4 * m + r] = (k < a.D) ? load_elem<XBF16>(a.x, rowc * a.ldx + k) : 0.f;
            }
    }

    int gt = 0;
    for (int q = 0; q < a.Q; ++q) {
        // ---- stage input: optional dump, ATen-order ||res||^2, B-operand layout ------------------
        if (a.resid_out && row_ok) {
            const int64_t ro = (row * a.Q + q) * (int64_t)a.D;
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                const int k0 = 8 * m + 4 * hi;
                if (VEC) {
                    if (XBF16) {
                        uint2 w;
                        w.x = (__float_as_uint(xr[4 * m + 0]) >> 16) | (__float_as_uint(xr[4 * m + 1]) & 0xffff0000u);
                        w.y = (__float_as_uint(xr[4 * m + 2]) >> 16) | (__float_as_uint(xr[4 * m + 3]) & 0xffff0000u);
                        *(uint2 *)((unsigned short *)a.resid_out + ro + k0) = w;
                    } else {
                        const f32x4 w = {xr[4 * m + 0], xr[4 * m + 1], xr[4 * m + 2], xr[4 * m + 3]};
                        *(f32x4 *)((float *)a.resid_out + ro + k0) = w;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (k0 + r < a.D) {
                            if (XBF16) ((unsigned short *)a.resid_out)[ro + k0 + r] = (unsigned short)(__float_as_uint(xr[4 * m + r]) >> 16);
                            else ((float *)a.resid_out)[ro + k0 + r] = xr[4 * m + r];
                        }
                }
            }
        }
        float x2;
        {
            f32x2 ch[4][2];
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) { ch[mm][0] = f32x2{0.f, 0.f}; ch[mm][1] = f32x2{0.f, 0.f}; }
#pragma unroll
            for (int m = 0; m < NG; ++m) {
                const f32x2 v01 = {xr[4 * m + 0], xr[4 * m + 1]}, v23 = {xr[4 * m + 2], xr[4 * m + 3]};
                ch[m & 3][0] = ch[m & 3][0] + v01 * v01;
                ch[m & 3][1] = ch[m & 3][1] + v23 * v23;
            }
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = ((ch[0][r >> 1][r & 1] + ch[1][r >> 1][r & 1]) + ch[2][r >> 1][r & 1]) + ch[3][r >> 1][r & 1];
            const float f_lo = ((p[0] + p[1]) + p[2]) + p[3];
            const float f_from_lo = __shfl(f_lo, j, 64);
            const float f_hi = (((f_from_lo + p[0]) + p[1]) + p[2]) + p[3];
            x2 = __shfl(f_hi, j + 32, 64);
        }
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            swap32(xr[4 * m + 0], xr[4 * m + 1]);
            swap32(xr[4 * m + 2], xr[4 * m + 3]);
        }

        // ---- sweep stage q's codebook --------------------------------------------------------------
        float bd = INFINITY, bs = INFINITY;
        int bi = 0;
        for (int ct = 0; ct < nt; ++ct, ++gt) {
            const int buf = gt & 1;
            __syncthreads();

            const char *tile = smem + buf * TILE_B;
            const f32x4 *ap = (const f32x4 *)tile + (hi * 32 + j);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const bool more = gt + 1 < total_tiles;
            mfma_sweep_tile<DT>(ap, xr, acc, tile_src(more ? gt + 1 : gt), smem + ((gt + 1) & 1) * TILE_B + piece_off, more ? my_pieces : 0);
            argmin_tile<0>(acc, (const float *)tile + 32 * DT, x2, ct * 32, hi, a.C, bd, bs, bi);
        }
        {
            const float od = __shfl_xor(bd, 32, 64);
            const int oi = __shfl_xor(bi, 32, 64);
            const bool take = (od < bd) || (od == bd && oi < bi);
            bd = take ? od : bd;
            bi = take ? oi : bi;
        }
        if (row_ok && hi == 0) a.idx_out[row * a.Q + q] = live ? (int64_t)bi : (int64_t)-1;

        // ---- residual update in the load layout --------------------------------------------------
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            swap32(xr[4 * m + 0], xr[4 * m + 1]);
            swap32(xr[4 * m + 2], xr[4 * m + 3]);
        }
        const float *er = a.embed + (size_t)q * a.embed_qstride + (size_t)bi * a.D;
        const unsigned short *erb = (const unsigned short *)((const char *)(a.packed + (size_t)q * a.packed_qstride) + a.bf16_off) + (size_t)bi * a.D;
        float lsum = 0.f;
        [[maybe_unused]] float lsum4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NG; ++m) {
            const int k0 = 8 * m + 4 * hi;
            float g[4];
            if (VEC && XBF16) {     // quantized is a bf16 tensor in the reference: pre-rounded rows
                const uint2 w = *(const uint2 *)(erb + k0);
                g[0] = __uint_as_float(w.x << 16); g[1] = __uint_as_float(w.x & 0xffff0000u);
                g[2] = __uint_as_float(w.y << 16); g[3] = __uint_as_float(w.y & 0xffff0000u);
            } else if (VEC) {
                const f32x4 w = *(const f32x4 *)(er + k0);
                g[0] = w.x; g[1] = w.y; g[2] = w.z; g[3] = w.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    g[r] = (k0 + r < a.D) ? er[k0 + r] : 0.f;
                    if (XBF16) g[r] = round_to_bf16(g[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float nr = xr[4 * m + r] - g[r];               // residual - quantized (rvq.py:524)
                if (XBF16) {                                   // bf16 tensors in the reference
                    const float df = g[r] - xr[4 * m + r];
                    lsum += df * df;
                    nr = round_to_bf16(nr);
                } else {
                    // (g - x)^2 == (x - g)^2 bit for bit: one difference serves the loss and the update, four independent sums
                    // (the single chain next to a second subtraction cost the fp32 kernels 200 - 380 spilled registers)
                    lsum4[r] = __builtin_fmaf(nr, nr, lsum4[r]);
                }
                xr[4 * m + r] = live ? nr : xr[4 * m + r];
            }
        }
        if (a.sqerr_partial) {
            if (!XBF16) lsum = (lsum4[0] + lsum4[1]) + (lsum4[2] + lsum4[3]);
            double ds = live ? (double)lsum : 0.0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ds += __shfl_xor(ds, o, 64);
            // per-wave partials go straight to global: [q][block][wave]
            if (lane == 0) a.sqerr_partial[((size_t)q * gridDim.x + blockIdx.x) * 4 + wave] = ds;
        }
    }
}

template <int DT, bool XBF16>
static int launch_rvq(const RvqArgs &a, hipStream_t st)
{
    constexpr int SMEM = 2 * (32 * DT + 256) * 4;
    static VqAttrOnce once_v, once_s;
    if (a.x_vec) {
        if (int rc = vq_set_max_smem(once_v, (const void *)vq_rvq_kernel<DT, XBF16, true>, SMEM, "vq_rvq_kernel")) return rc;
        hipLaunchKernelGGL((vq_rvq_kernel<DT, XBF16, true>), dim3((unsigned)vqhip_assign_blocks(a.N)), dim3(256), SMEM, st, a);
    } else {
        if (int rc = vq_set_max_smem(once_s, (const void *)vq_rvq_kernel<DT, XBF16, false>, SMEM, "vq_rvq_kernel")) return rc;
        hipLaunchKernelGGL((vq_rvq_kernel<DT, XBF16, false>), dim3((unsigned)vqhip_assign_blocks(a.N)), dim3(256), SMEM, st, a);
    }
    return launch_status("vq_rvq_kernel");
}

extern "C" int vqhip_rvq_forward(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                                 const float *packed, int64_t packed_qstride, const float *embed, int64_t embed_qstride,
                                 int C, int Q, int64_t *idx_out, void *resid_out, double *sqerr_partial,
                                 const uint8_t *row_mask, void *stream)
{
    if (N < 0 || C <= 0 || Q < 1) VQ_FAIL(VQHIP_EINVAL, "rvq_forward: bad size");
    if (N == 0) return 0;
    if (!x || !packed || !embed || !idx_out) VQ_FAIL(VQHIP_EINVAL, "rvq_forward: null pointer");
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "rvq_forward: unknown x dtype %d", x_dtype);
    const int DT = pick_dt(D);
    if (D < 1 || DT == 0) VQ_FAIL(VQHIP_EDIM, "rvq_forward: D=%d unsupported (1..512)", D);
    if (D & 31) VQ_FAIL(VQHIP_EDIM, "rvq_forward: the fused residual loop needs D %% 32 == 0 (got %d); use per-stage vqhip_assign", D);
    if (ldx < D) VQ_FAIL(VQHIP_EINVAL, "rvq_forward: row stride smaller than D");
    if (((uintptr_t)packed) & 15) VQ_FAIL(VQHIP_EALIGN, "rvq_forward: packed must be 16-byte aligned");
    RvqArgs a;
    a.x = x; a.N = N; a.D = D; a.ldx = ldx; a.packed = packed; a.packed_qstride = packed_qstride; a.embed = embed;
    a.embed_qstride = embed_qstride; a.C = C; a.n_tiles = (C + 31) / 32; a.Q = Q; a.idx_out = idx_out;
    a.resid_out = resid_out; a.sqerr_partial = sqerr_partial; a.row_mask = row_mask;
    a.bf16_off = packed_bf16_offset(C, D);
    const int xes = (x_dtype == VQHIP_BF16) ? 2 : 4;
    a.x_vec = (D == DT) && (((uintptr_t)x) % (4 * xes) == 0) && ((ldx * xes) % (4 * xes) == 0) && ((((uintptr_t)embed) & 15) == 0) &&
              (!resid_out || (((uintptr_t)resid_out) % (4 * xes) == 0)) && ((embed_qstride % 4) == 0) && ((packed_qstride % 4) == 0);
    hipStream_t st = (hipStream_t)stream;
    const bool bf = (x_dtype == VQHIP_BF16);
    switch (DT) {
        case 32: return bf ? launch_rvq<32, true>(a, st) : launch_rvq<32, false>(a, st);
        case 64: return bf ? launch_rvq<64, true>(a, st) : launch_rvq<64, false>(a, st);
        case 128: return bf ? launch_rvq<128, true>(a, st) : launch_rvq<128, false>(a, st);
        case 256: return bf ? launch_rvq<256, true>(a, st) : launch_rvq<256, false>(a, st);
        default: return bf ? launch_rvq<512, true>(a, st) : launch_rvq<512, false>(a, st);
    }
}

// ------------------------------------------------------------------------------------------------
// gradient routing through the quantizer (one wave per row, rows independent)
//   mode 1  straight-through (vqp.py:282-283):  out = x + (q - x);            d out / d x = I
//   mode 2  rotation trick (vqp.py:287-318, arXiv:2410.06424 s4.2), u, qh, w, s all detached:
//           out = s (e - 2 (e.w) w + 2 (e.u) qh),   u = e/|e|, qh = q/|q|, w = l2norm(u + qh), s = |q|/|e|
//           grad_e = s (g - 2 (g.w) w + 2 (g.qh) u)
//   commit loss mean((q.detach() - x)^2) (vqp.py:1327): grad_x += coef * 2 (x - q), coef = dL/d(sum of squares),
//   a DEVICE scalar (no host sync), rows with row_mask == 0 excluded.
// ------------------------------------------------------------------------------------------------
// (row reductions and the routed value itself: vq_route_math.h, shared with the screening kernel's chain prologue)

// Rows per wave of the routing kernels.  One wave per row (LPR = 64) spends most of its instructions on the five row reductions
// of a rotation-trick stage (4 DPP + 4 v_readlane + 3 adds each) for 4 elements per lane -- and half its lanes idle at D = 128.
// With one row per 16 lanes (D <= 256: 4 .. 16 elements per lane) a reduction is 4 DPP adds for FOUR rows at once: the kernels
// go from instruction-bound towards their HBM traffic (forward + backward of a step: cfg 3 1.5 -> 0.7 ms, cfg 5 6.1 -> 1.4 ms).
//
// Row access of the routing kernels: lane l owns the elements 4 (LPR k + l) + i, k < NE / 4, i < 4 -- four CONTIGUOUS elements per
// slice (NE = 4: D <= 256, one slice; NE = 8: two), moved by one 8-byte (bf16) / 16-byte (fp32) access when the row allows it (`vec`: D % 4 == 0 and base / stride aligned
// to 4 elements); otherwise element by element.  (The first version gave lane l the elements l + 64 k: 2-byte accesses for bf16.)
template <bool BF16, int NE, int LPR = 64>
__device__ __forceinline__ void row_load8(const void *base, int64_t off, int D, int lane, bool vec, float (&v)[NE])
{
#pragma unroll
    for (int k = 0; k < NE / 4; ++k) {
        const int d0 = 4 * (LPR * k + lane);
        if (vec && d0 < D) {
            if (BF16) {
                const uint2 w = *(const uint2 *)((const unsigned short *)base + off + d0);
                v[4 * k + 0] = __uint_as_float(w.x << 16); v[4 * k + 1] = __uint_as_float(w.x & 0xffff0000u);
                v[4 * k + 2] = __uint_as_float(w.y << 16); v[4 * k + 3] = __uint_as_float(w.y & 0xffff0000u);
            } else {
                const f32x4 w = *(const f32x4 *)((const float *)base + off + d0);
                v[4 * k + 0] = w.x; v[4 * k + 1] = w.y; v[4 * k + 2] = w.z; v[4 * k + 3] = w.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * k + i] = (d0 + i < D) ? load_elem<BF16>(base, off + d0 + i) : 0.f;
        }
    }
}

template <bool BF16, int NE, int LPR = 64>
__device__ __forceinline__ void row_store8(void *base, int64_t off, int D, int lane, bool vec, const float (&v)[NE])
{
#pragma unroll
    for (int k = 0; k < NE / 4; ++k) {
        const int d0 = 4 * (LPR * k + lane);
        if (vec && d0 < D) {
            if (BF16) {
                uint2 w;
                w.x = (unsigned)f32_to_bf16_rne(v[4 * k + 0]) | ((unsigned)f32_to_bf16_rne(v[4 * k + 1]) << 16);
                w.y = (unsigned)f32_to_bf16_rne(v[4 * k + 2]) | ((unsigned)f32_to_bf16_rne(v[4 * k + 3]) << 16);
                *(uint2 *)((unsigned short *)base + off + d0) = w;
            } else {
                *(f32x4 *)((float *)base + off + d0) = f32x4{v[4 * k + 0], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (d0 + i < D) {
                    if (BF16) ((unsigned short *)base)[off + d0 + i] = f32_to_bf16_rne(v[4 * k + i]);
                    else ((float *)base)[off + d0 + i] = v[4 * k + i];
                }
        }
    }
}

static inline bool rows_vec4(const void *p, int64_t ld, int D, int es)
{
    return p == nullptr || ((D & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)p) % (size_t)(4 * es)) == 0);
}

struct RouteArgs {
    const void *x;
    const void *q;
    const void *g;          // backward only, nullable
    void *out;              // forward: out; backward: grad_x
    int64_t N;
    int D;
    int64_t ldx, ldq, ldg, ldo;
    const float *loss_coef; // backward only, nullable device scalar
    const uint8_t *row_mask;
    int mode;
    int vec;                // every row pointer / stride allows 4-element accesses
    // q rows gathered from a code table instead of read from an [N, D] tensor: q[n] = a.q[qidx[n * qidx_stride]] (a.q = [C, D] in the
    // rows' dtype, ldq = D) -- the q tensor then never has to exist (vqhip_route_fwd_gather / _bwd_gather); sub: out = x - route(x, q),
    // the residual step of a residual VQ whose layers return the ROUTED value (vqhip_route_residual, rvq.py:524)
    const int64_t *qidx;
    int64_t qidx_stride;
    int sub;
    // backward: what the forward left on the rows with row_mask == 0 (vqhip_mask_fill_rows, the reference's torch.where(mask, quantize,
    // orig_input | zeros) of vqp.py:1386-1394): 0 = the routed value like any row, 1 = x itself (the upstream gradient passes through),
    // 2 = zeros (no gradient)
    int masked_rows;
};

template <bool BF16, bool BWD, int NE, int LPR>
__global__ void __launch_bounds__(256) vq_route_kernel(const RouteArgs a)
{
    constexpr int RPW = 64 / LPR;                      // rows per wave
    const int lane = threadIdx.x & (LPR - 1);          // lane within its row
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LPR);
    const bool valid = n0 < a.N;
    if (LPR == 64 && !valid) return;
    const int64_t n = valid ? n0 : a.N - 1;            // (rows share a wave: the ones past the end repeat the last row and store nothing)
    float e[NE], qv[NE], g[NE];
    row_load8<BF16, NE, LPR>(a.x, n * a.ldx, a.D, lane, a.vec != 0, e);
    const int64_t qrow = a.qidx ? max((int64_t)0, a.qidx[n * a.qidx_stride]) : n;   // (-1: a padding row, its value is not used)
    row_load8<BF16, NE, LPR>(a.q, qrow * a.ldq, a.D, lane, a.vec != 0, qv);
    if (BWD && a.g) row_load8<BF16, NE, LPR>(a.g, n * a.ldg, a.D, lane, a.vec != 0, g);
    float r[NE];
    if (!BWD) {
        vq_route_value<NE, LPR, BF16>(e, qv, a.mode, r);            // vq_route_math.h (mode 1 / 2)
        if (a.sub) {                                                  // the next stage's input: residual - quantized.detach() (rvq.py:524);
#pragma unroll
            for (int k = 0; k < NE; ++k) r[k] = e[k] - (BF16 ? round_to_bf16(r[k]) : r[k]);   // bf16: `quantized` is a bf16 tensor
        }
    } else if (a.mode == 2 && a.g) {
        float u[NE], qh[NE], w[NE], sc;
        vq_rot_frame<NE, LPR, BF16>(e, qv, u, qh, w, sc);             // (bf16: the frame the reference's graph saved, rounded op by op)
        vq_rot_bwd<NE, LPR>(g, u, qh, w, sc, r);
    } else {
#pragma unroll
        for (int k = 0; k < NE; ++k) r[k] = a.g ? g[k] : 0.f;
    }
    if (BWD && a.loss_coef) {
        const bool counted = !a.row_mask || a.row_mask[n] != 0;
        const float c2 = counted ? 2.f * (*a.loss_coef) : 0.f;
#pragma unroll
        for (int k = 0; k < NE; ++k) r[k] += c2 * (e[k] - qv[k]);
    }
    if (BWD && a.masked_rows && a.row_mask && a.row_mask[n] == 0) {
#pragma unroll
        for (int k = 0; k < NE; ++k) r[k] = (a.masked_rows == 1 && a.g) ? g[k] : 0.f;
    }
    if (valid) row_store8<BF16, NE, LPR>(a.out, n * a.ldo, a.D, lane, a.vec != 0, r);
}

static int route_launch(const RouteArgs &a, int dtype, bool bwd, hipStream_t st)
{
    if (a.N == 0) return 0;
    dim3 grid((unsigned)((a.N + 3) / 4)), grid16((unsigned)((a.N + 15) / 16));
    // D <= 256: one row per 16 lanes (NE = 4 ceil(D / 64) elements per lane), else one row per wave
#define VQ_ROUTE_LAUNCH(BF, BW) do { if (a.D <= 64) hipLaunchKernelGGL((vq_route_kernel<BF, BW, 4, 16>), grid16, dim3(256), 0, st, a); \
                                     else if (a.D <= 128) hipLaunchKernelGGL((vq_route_kernel<BF, BW, 8, 16>), grid16, dim3(256), 0, st, a); \
                                     else if (a.D <= 256) hipLaunchKernelGGL((vq_route_kernel<BF, BW, 16, 16>), grid16, dim3(256), 0, st, a); \
                                     else if (a.D <= 512) hipLaunchKernelGGL((vq_route_kernel<BF, BW, 8, 64>), grid, dim3(256), 0, st, a); \
                                     else if (a.D <= 1024) hipLaunchKernelGGL((vq_route_kernel<BF, BW, 16, 64>), grid, dim3(256), 0, st, a); \
                                     else hipLaunchKernelGGL((vq_route_kernel<BF, BW, 32, 64>), grid, dim3(256), 0, st, a); } while (0)
    if (dtype == VQHIP_BF16) {
        if (bwd) VQ_ROUTE_LAUNCH(true, true); else VQ_ROUTE_LAUNCH(true, false);
    } else {
        if (bwd) VQ_ROUTE_LAUNCH(false, true); else VQ_ROUTE_LAUNCH(false, false);
    }
#undef VQ_ROUTE_LAUNCH
    return launch_status("vq_route_kernel");
}

extern "C" int vqhip_route_fwd(const void *x, const void *q, int dtype, int64_t N, int D, int64_t ldx, int64_t ldq,
                               void *out, int64_t ldo, int mode, void *stream)
{
    if (N < 0 || !x || !q || !out) VQ_FAIL(VQHIP_EINVAL, "route_fwd: bad argument");
    if (D < 1 || D > VQ_WIDE_MAX_D) VQ_FAIL(VQHIP_EDIM, "route_fwd: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (mode != 1 && mode != 2) VQ_FAIL(VQHIP_EINVAL, "route_fwd: mode must be 1 (straight-through) or 2 (rotation trick)");
    if (dtype != VQHIP_F32 && dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "route_fwd: unknown dtype");
    RouteArgs a;
    a.x = x; a.q = q; a.g = nullptr; a.out = out; a.N = N; a.D = D; a.ldx = ldx; a.ldq = ldq; a.ldg = 0; a.ldo = ldo;
    a.loss_coef = nullptr; a.row_mask = nullptr; a.mode = mode; a.qidx = nullptr; a.qidx_stride = 0; a.sub = 0; a.masked_rows = 0;
    const int es = dtype == VQHIP_BF16 ? 2 : 4;
    a.vec = rows_vec4(x, ldx, D, es) && rows_vec4(q, ldq, D, es) && rows_vec4(out, ldo, D, es);
    return route_launch(a, dtype, false, (hipStream_t)stream);
}

// The residual a ResidualVQ stage hands to the next one when its layer returned the ROUTED value (training with an input that
// requires grad: `residual = residual - quantized.detach()`, rvq.py:524 with vqp.py:1225-1233): out = x - route(x, embed[idx]),
// the arithmetic of vqhip_route_fwd / vqhip_rvq_route bit for bit (vq_route_math.h; bf16 rows: the routed value rounded to bf16 as
// the tensor the layer returned, then the difference); codes [C, D] in the rows' dtype.  An HBM-bound kernel of its own at
// full occupancy: inside the screening kernel's prologue (round 4's first form) the row reductions ran on 256-register waves and
// doubled that kernel's time (459 vs 217 us per cfg-3 stage).
extern "C" int vqhip_route_residual(const void *x, int dtype, int64_t N, int D, int64_t ldx, const void *codes, const int64_t *idx,
                                    int64_t idx_stride, int mode, void *out, int64_t ldo, void *stream)
{
    if (!x || !codes || !idx || !out) VQ_FAIL(VQHIP_EINVAL, "route_residual: null pointer");
    if (dtype != VQHIP_F32 && dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "route_residual: unknown dtype");
    if (N < 0 || D < 1 || D > VQ_WIDE_MAX_D || idx_stride < 1) VQ_FAIL(VQHIP_EINVAL, "route_residual: bad size");
    if (mode != 1 && mode != 2) VQ_FAIL(VQHIP_EINVAL, "route_residual: mode must be 1 (straight-through) or 2 (rotation trick)");
    RouteArgs a;
    a.x = x; a.q = codes; a.g = nullptr; a.out = out; a.N = N; a.D = D; a.ldx = ldx; a.ldq = D; a.ldg = 0; a.ldo = ldo;
    a.loss_coef = nullptr; a.row_mask = nullptr; a.mode = mode; a.qidx = idx; a.qidx_stride = idx_stride; a.sub = 1; a.masked_rows = 0;
    const int es = dtype == VQHIP_BF16 ? 2 : 4;
    a.vec = rows_vec4(x, ldx, D, es) && rows_vec4(codes, D, D, es) && rows_vec4(out, ldo, D, es);
    return route_launch(a, dtype, false, (hipStream_t)stream);
}

extern "C" int vqhip_route_bwd(const void *x, const void *q, const void *g_out, int dtype, int64_t N, int D,
                               int64_t ldx, int64_t ldq, int64_t ldg, const float *loss_coef, const uint8_t *row_mask,
                               int mode, int masked_rows, void *grad_x, int64_t ldo, void *stream)
{
    if (masked_rows < 0 || masked_rows > 2) VQ_FAIL(VQHIP_EINVAL, "route_bwd: masked_rows must be 0, 1 or 2");
    if (N < 0 || !x || !q || !grad_x) VQ_FAIL(VQHIP_EINVAL, "route_bwd: bad argument");
    if (D < 1 || D > VQ_WIDE_MAX_D) VQ_FAIL(VQHIP_EDIM, "route_bwd: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (mode < 0 || mode > 2) VQ_FAIL(VQHIP_EINVAL, "route_bwd: mode must be 0 (loss only), 1 or 2");
    if (dtype != VQHIP_F32 && dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "route_bwd: unknown dtype");
    RouteArgs a;
    a.x = x; a.q = q; a.g = (mode == 0) ? nullptr : g_out; a.out = grad_x; a.N = N; a.D = D; a.ldx = ldx; a.ldq = ldq;
    a.ldg = ldg; a.ldo = ldo; a.loss_coef = loss_coef; a.row_mask = row_mask; a.mode = (mode == 0) ? 1 : mode;
    a.qidx = nullptr; a.qidx_stride = 0; a.sub = 0; a.masked_rows = masked_rows;
    const int es = dtype == VQHIP_BF16 ? 2 : 4;
    a.vec = rows_vec4(x, ldx, D, es) && rows_vec4(q, ldq, D, es) && rows_vec4(a.g, ldg, D, es) && rows_vec4(grad_x, ldo, D, es);
    return route_launch(a, dtype, true, (hipStream_t)stream);
}

// The same two kernels with q GATHERED from a code table by index (codes [C, D] in the rows' dtype, contiguous: the bf16 copy inside
// the packed codebook for bf16 rows, embed for fp32 rows): a training step whose input requires grad then neither writes nor re-reads
// an [N, D] q tensor -- the search returns indices only, the forward value and the gradient gather the code rows from L2.
extern "C" int vqhip_route_fwd_gather(const void *x, const void *codes, const int64_t *idx, int64_t idx_stride, int dtype, int64_t N, int D,
                                      int64_t ldx, void *out, int64_t ldo, int mode, void *stream)
{
    if (!x || !codes || !idx || !out) VQ_FAIL(VQHIP_EINVAL, "route_fwd_gather: null pointer");
    if (dtype != VQHIP_F32 && dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "route_fwd_gather: unknown dtype");
    if (N < 0 || D < 1 || D > VQ_WIDE_MAX_D || idx_stride < 1) VQ_FAIL(VQHIP_EINVAL, "route_fwd_gather: bad size");
    if (mode != 1 && mode != 2) VQ_FAIL(VQHIP_EINVAL, "route_fwd_gather: mode must be 1 (straight-through) or 2 (rotation trick)");
    RouteArgs a;
    a.x = x; a.q = codes; a.g = nullptr; a.out = out; a.N = N; a.D = D; a.ldx = ldx; a.ldq = D; a.ldg = 0; a.ldo = ldo;
    a.loss_coef = nullptr; a.row_mask = nullptr; a.mode = mode; a.qidx = idx; a.qidx_stride = idx_stride; a.sub = 0; a.masked_rows = 0;
    const int es = dtype == VQHIP_BF16 ? 2 : 4;
    a.vec = rows_vec4(x, ldx, D, es) && rows_vec4(codes, D, D, es) && rows_vec4(out, ldo, D, es);
    return route_launch(a, dtype, false, (hipStream_t)stream);
}

extern "C" int vqhip_route_bwd_gather(const void *x, const void *codes, const int64_t *idx, int64_t idx_stride, const void *g_out, int dtype,
                                      int64_t N, int D, int64_t ldx, int64_t ldg, const float *loss_coef, const uint8_t *row_mask,
                                      int mode, int masked_rows, void *grad_x, int64_t ldo, void *stream)
{
    if (masked_rows < 0 || masked_rows > 2) VQ_FAIL(VQHIP_EINVAL, "route_bwd_gather: masked_rows must be 0, 1 or 2");
    if (!x || !codes || !idx || !grad_x) VQ_FAIL(VQHIP_EINVAL, "route_bwd_gather: null pointer");
    if (dtype != VQHIP_F32 && dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "route_bwd_gather: unknown dtype");
    if (N < 0 || D < 1 || D > VQ_WIDE_MAX_D || idx_stride < 1) VQ_FAIL(VQHIP_EINVAL, "route_bwd_gather: bad size");
    if (mode < 0 || mode > 2) VQ_FAIL(VQHIP_EINVAL, "route_bwd_gather: mode must be 0, 1 or 2");
    RouteArgs a;
    a.x = x; a.q = codes; a.g = (mode == 0) ? nullptr : g_out; a.out = grad_x; a.N = N; a.D = D; a.ldx = ldx; a.ldq = D;
    a.ldg = ldg; a.ldo = ldo; a.loss_coef = loss_coef; a.row_mask = row_mask; a.mode = (mode == 0) ? 1 : mode;
    a.qidx = idx; a.qidx_stride = idx_stride; a.sub = 0; a.masked_rows = masked_rows;
    const int es = dtype == VQHIP_BF16 ? 2 : 4;
    a.vec = rows_vec4(x, ldx, D, es) && rows_vec4(codes, D, D, es) && rows_vec4(a.g, ldg, D, es) && rows_vec4(grad_x, ldo, D, es);
    return route_launch(a, dtype, true, (hipStream_t)stream);
}

// Padding rows of a masked batch (vqp.py:1386-1394: quantize = where(mask, quantize, orig_input | zeros), indices = where(mask, indices,
// -1)): in place on the rows with row_mask == 0 only -- the traffic of the padding, not three passes over N x D.  16 lanes per row.
template <bool BF16>
__global__ void __launch_bounds__(256) vq_mask_fill_kernel(void *q, const void *x, int64_t N, int D, int64_t ldq, int64_t ldx,
                                                           const uint8_t *row_mask, int64_t *idx, int64_t idx_stride, int zeros, int vec)
{
    const int l16 = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= N || row_mask[row] != 0) return;
    if (idx && l16 == 0) idx[row * idx_stride] = -1;
    if (!q) return;
    for (int d = 4 * l16; d < D; d += 64) {
        if (vec) {
            if (BF16) {
                uint2 w = {0u, 0u};
                if (!zeros) w = *(const uint2 *)((const unsigned short *)x + row * ldx + d);
                *(uint2 *)((unsigned short *)q + row * ldq + d) = w;
            } else {
                f32x4 w = {0.f, 0.f, 0.f, 0.f};
                if (!zeros) w = *(const f32x4 *)((const float *)x + row * ldx + d);
                *(f32x4 *)((float *)q + row * ldq + d) = w;
            }
        } else {
            for (int i = 0; i < 4 && d + i < D; ++i) {
                if (BF16) ((unsigned short *)q)[row * ldq + d + i] = zeros ? (unsigned short)0 : ((const unsigned short *)x)[row * ldx + d + i];
                else ((float *)q)[row * ldq + d + i] = zeros ? 0.f : ((const float *)x)[row * ldx + d + i];
            }
        }
    }
}

extern "C" int vqhip_mask_fill_rows(void *q, const void *x, int dtype, int64_t N, int D, int64_t ldq, int64_t ldx, const uint8_t *row_mask,
                                    int64_t *idx, int64_t idx_stride, int zeros, void *stream)
{
    if (N < 0 || !row_mask || (!q && !idx)) VQ_FAIL(VQHIP_EINVAL, "mask_fill_rows: bad argument");
    if (N == 0) return 0;
    if (dtype != VQHIP_F32 && dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "mask_fill_rows: unknown dtype");
    if (q && (D < 1 || ldq < D || (!zeros && (!x || ldx < D)))) VQ_FAIL(VQHIP_EINVAL, "mask_fill_rows: bad rows");
    if (idx && idx_stride < 1) VQ_FAIL(VQHIP_EINVAL, "mask_fill_rows: idx_stride < 1");
    const int es = dtype == VQHIP_BF16 ? 2 : 4;
    const int vec = (q && rows_vec4(q, ldq, D, es) && (zeros || rows_vec4(x, ldx, D, es))) ? 1 : 0;
    const unsigned blocks = (unsigned)((N + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == VQHIP_BF16) hipLaunchKernelGGL(vq_mask_fill_kernel<true>, dim3(blocks), dim3(256), 0, st, q, x, N, D, ldq, ldx, row_mask, idx, idx_stride, zeros, vec);
    else hipLaunchKernelGGL(vq_mask_fill_kernel<false>, dim3(blocks), dim3(256), 0, st, q, x, N, D, ldq, ldx, row_mask, idx, idx_stride, zeros, vec);
    return launch_status("vq_mask_fill_kernel");
}

// ------------------------------------------------------------------------------------------------
// gradient routing through the residual loop (ResidualVQ.forward, rvq.py:469-568 with quant_grad_frac = 0): stage q sees the
// residual r_q = r_{q-1} - c_{q-1} (detached subtraction, rvq.py:524), its routed output is added to quantized_out (rvq.py:525),
// its commitment loss compares r_q with c_q.  dL/dx therefore sums, over the stages, the straight-through / rotation-trick
// Jacobian of that stage applied to the upstream gradient plus the stage's commit-loss gradient (d r_q / d x = I).
// One wave per row keeps r in registers, gathers the Q code rows (L2) and never materialises a per-stage tensor:
//   FWD: out = sum_q route_fwd(r_q, c_q)       BWD: out = sum_q J_q^T g + 2 coef_q (r_q - c_q)
// bf16 tensors: every tensor op of the reference rounds to bf16 (code rows, residual, routed value, running sum).
// ------------------------------------------------------------------------------------------------
struct RvqRouteArgs {
    const void *x, *g;
    void *out;
    const float *embed;        // [Q or 1][C, D] fp32
    int64_t qstride;           // elements between the codebooks of consecutive stages (0: shared)
    const int64_t *idx;        // [N, idx_stride]
    int64_t idx_stride;
    const float *loss_coef;    // BWD, nullable: [Q] device floats, d loss_total / d (sum of squares of stage q)
    const uint8_t *row_mask;
    int64_t N, ldx, ldg, ldo;
    int D, Q, mode, vec;
    int resid_routed;          // the residual loop subtracted the routed value (mode != 0), not the code row, between stages
};

template <bool BF16, bool BWD, int NE, int LPR>
__global__ void __launch_bounds__(256) vq_rvq_route_kernel(const RvqRouteArgs a)
{
    constexpr int RPW = 64 / LPR;                      // rows per wave
    const int lane = threadIdx.x & (LPR - 1);          // lane within its row
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LPR);
    const bool valid = n0 < a.N;
    if (LPR == 64 && !valid) return;
    const int64_t n = valid ? n0 : a.N - 1;
    float r[NE], g[NE], acc[NE];
    row_load8<BF16, NE, LPR>(a.x, n * a.ldx, a.D, lane, a.vec != 0, r);
    if (BWD && a.g) row_load8<BF16, NE, LPR>(a.g, n * a.ldg, a.D, lane, a.vec != 0, g);
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        acc[k] = 0.f;
        if (!(BWD && a.g)) g[k] = 0.f;
    }
    const bool counted = !a.row_mask || a.row_mask[n] != 0;
    bool alive = true;
    for (int q = 0; q < a.Q; ++q) {
        const int64_t code0 = a.idx[n * a.idx_stride + q];
        alive = alive && code0 >= 0;                             // dropped-out quantizers (rvq.py:478-482) and masked rows: this row is done
        if (!__any(alive)) break;                                // (the rows of a wave end independently; a finished row idles along)
        const int64_t code = alive ? code0 : 0;
        const float *cp = a.embed + (size_t)q * a.qstride + (size_t)code * a.D;
        float c[NE];
        row_load8<false, NE, LPR>(cp, 0, a.D, lane, (a.D & 3) == 0, c);   // code rows: fp32, [C, D] contiguous, 16-byte aligned when D % 4 == 0
        if (BF16) {
#pragma unroll
            for (int k = 0; k < NE; ++k) c[k] = round_to_bf16(c[k]);
        }
        // tf: the value the layer returns for this row (vq_route_math.h) -- what the forward sums, and with resid_routed what
        // rvq.py:524 subtracts from the residual.  t: this stage's term of the output (forward) / of dL/dx (backward).
        float t[NE], tf[NE];
        const bool need_tf = !BWD || (a.resid_routed && a.mode != 0);
        if (a.mode == 2) {                                       // rotation trick
            float u[NE], qh[NE], w[NE], sc;
            vq_rot_frame<NE, LPR, BF16>(r, c, u, qh, w, sc);
            if (need_tf) vq_rot_fwd<NE, LPR, BF16>(r, u, qh, w, sc, tf);
            if (BWD) vq_rot_bwd<NE, LPR>(g, u, qh, w, sc, t);
        } else {
            if (need_tf) vq_route_value<NE, LPR, BF16>(r, c, a.mode, tf);
            if (BWD) {
#pragma unroll
                for (int k = 0; k < NE; ++k) t[k] = a.mode == 1 ? g[k] : 0.f;
            }
        }
        if (!BWD) {
#pragma unroll
            for (int k = 0; k < NE; ++k) t[k] = tf[k];
        }
        if (BWD && a.loss_coef && counted) {
            const float c2 = 2.f * a.loss_coef[q];
#pragma unroll
            for (int k = 0; k < NE; ++k) t[k] += c2 * (r[k] - c[k]);
        }
        if (alive) {
            // the next stage's input: residual - quantized.detach() (rvq.py:524).  `quantized` is the ROUTED value when the layer
            // routed gradients to its input (training, input requires grad: vqp.py:1225-1233), else the code row itself;
            // bf16 tensors round after every tensor op.
            const bool routed = a.resid_routed && a.mode != 0;
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const float sub = routed ? (BF16 ? round_to_bf16(tf[k]) : tf[k]) : c[k];
                if (BF16 && !BWD) acc[k] = round_to_bf16(acc[k] + round_to_bf16(t[k]));
                else              acc[k] += t[k];
                r[k] = BF16 ? round_to_bf16(r[k] - sub) : (r[k] - sub);
            }
        }
    }
    if (valid) row_store8<BF16, NE, LPR>(a.out, n * a.ldo, a.D, lane, a.vec != 0, acc);
}

extern "C" int vqhip_rvq_route(const void *x, int dtype, int64_t N, int D, int64_t ldx, const float *embed, int64_t embed_qstride,
                               int C, const int64_t *idx, int64_t idx_stride, int Q, int mode, int resid_routed, const void *g_out, int64_t ldg,
                               const float *loss_coef, const uint8_t *row_mask, int backward, void *out, int64_t ldo, void *stream)
{
    if (N < 0 || !x || !embed || !idx || !out || C <= 0) VQ_FAIL(VQHIP_EINVAL, "rvq_route: bad argument");
    if (D < 1 || D > 512) VQ_FAIL(VQHIP_EDIM, "rvq_route: D=%d unsupported (1..512)", D);
    if (Q < 1 || idx_stride < Q) VQ_FAIL(VQHIP_EINVAL, "rvq_route: Q=%d, idx_stride=%lld", Q, (long long)idx_stride);
    if (mode < 0 || mode > 2) VQ_FAIL(VQHIP_EINVAL, "rvq_route: mode must be 0 (plain sum / loss only), 1 or 2");
    if (dtype != VQHIP_F32 && dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "rvq_route: unknown dtype");
    if (N == 0) return 0;
    RvqRouteArgs a;
    a.x = x; a.g = (backward && mode != 0) ? g_out : nullptr; a.out = out; a.embed = embed; a.qstride = embed_qstride;
    a.idx = idx; a.idx_stride = idx_stride; a.loss_coef = backward ? loss_coef : nullptr; a.row_mask = row_mask;
    a.N = N; a.ldx = ldx; a.ldg = ldg; a.ldo = ldo; a.D = D; a.Q = Q; a.mode = mode; a.resid_routed = resid_routed;
    const int es = dtype == VQHIP_BF16 ? 2 : 4;
    a.vec = rows_vec4(x, ldx, D, es) && rows_vec4(a.g, ldg, D, es) && rows_vec4(out, ldo, D, es);
    if ((D & 3) == 0 && ((((uintptr_t)embed) & 15) || (embed_qstride & 3))) VQ_FAIL(VQHIP_EALIGN, "rvq_route: embed must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((N + 3) / 4)), grid16((unsigned)((N + 15) / 16));
#define VQ_RVQ_ROUTE_LAUNCH(BF, BW) do { if (D <= 64) hipLaunchKernelGGL((vq_rvq_route_kernel<BF, BW, 4, 16>), grid16, dim3(256), 0, st, a); \
                                         else if (D <= 128) hipLaunchKernelGGL((vq_rvq_route_kernel<BF, BW, 8, 16>), grid16, dim3(256), 0, st, a); \
                                         else if (D <= 256) hipLaunchKernelGGL((vq_rvq_route_kernel<BF, BW, 16, 16>), grid16, dim3(256), 0, st, a); \
                                         else hipLaunchKernelGGL((vq_rvq_route_kernel<BF, BW, 8, 64>), grid, dim3(256), 0, st, a); } while (0)
    if (dtype == VQHIP_BF16) {
        if (backward) VQ_RVQ_ROUTE_LAUNCH(true, true); else VQ_RVQ_ROUTE_LAUNCH(true, false);
    } else {
        if (backward) VQ_RVQ_ROUTE_LAUNCH(false, true); else VQ_RVQ_ROUTE_LAUNCH(false, false);
    }
#undef VQ_RVQ_ROUTE_LAUNCH
    return launch_status("vq_rvq_route_kernel");
}

// ------------------------------------------------------------------------------------------------
// partial reduction (commit loss)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vq_reduce_kernel(const double *__restrict__ p, int64_t n, double scale, float *out)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += p[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = (float)(red[0] * scale);
}

extern "C" int vqhip_reduce_partials(const double *partials, int64_t n, double scale, float *out, void *stream)
{
    if (!out || n < 0 || (n > 0 && !partials)) VQ_FAIL(VQHIP_EINVAL, "reduce_partials: bad argument");
    hipLaunchKernelGGL(vq_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n, scale, out);
    return launch_status("vq_reduce_kernel");
}

// R rows of partials in one launch (the per-stage losses of a residual VQ): out[r] = scale * sum(partials[r * stride .. + n))
__global__ void __launch_bounds__(256) vq_reduce_rows_kernel(const double *__restrict__ p, int64_t n, int64_t stride, double scale, float *out)
{
    __shared__ double red[256];
    const double *row = p + (size_t)blockIdx.x * stride;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += row[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] * scale);
}

extern "C" int vqhip_reduce_partials_rows(const double *partials, int R, int64_t n, int64_t stride, double scale, float *out, void *stream)
{
    if (!out || R < 1 || n < 0 || stride < n || (n > 0 && !partials)) VQ_FAIL(VQHIP_EINVAL, "reduce_partials_rows: bad argument");
    hipLaunchKernelGGL(vq_reduce_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, partials, n, stride, scale, out);
    return launch_status("vq_reduce_rows_kernel");
}

// ------------------------------------------------------------------------------------------------
// EMA sufficient statistics: counting sort of the rows by code, then full-row segmented sums.
//
//   vq_hist_kernel     count[c]           LDS integer histogram per workgroup -> global int atomics
//   vq_scan_kernel     segment offsets, chunk offsets (<= VQ_SEG_CH rows per work item), cursors
//   vq_scatter_kernel  perm[] = row ids grouped by code: LDS returning int atomics give the rank inside
//                      the workgroup, ONE global atomic per (workgroup, code) reserves the slots
//                      (cursors padded to one 64-byte line each so they spread over the L2 channels)
//   vq_segsum_kernel   one wave per (code, chunk): streams whole rows (512 B .. 2 KiB contiguous) with
//                      16 rows in flight, sums in registers, one global fp32 atomic per column
//
// Why not LDS-privatised fp32 accumulators (the first implementation): ds_add_f32 retires ~1 lane per
// 3 clocks on gfx950 -- measured 1.36 ms vs 0.36 ms for the same kernel with plain racy adds at
// N = 2^20, C = 1024, D = 256 (profiles/README.md).  Integer LDS atomics are only used for counting.
// ------------------------------------------------------------------------------------------------
#ifndef VQ_SEG_CH
#define VQ_SEG_CH 256          // rows per segmented-sum work item
#endif
#define VQ_HIST_LDS_MAX 16384  // codes whose histogram fits the LDS path (64 KiB)
#ifndef VQ_SORT_THREADS
#define VQ_SORT_THREADS 512    // threads of a counting-sort workgroup (LDS path), 16 rows each.  256 -> 512 (half the workgroups, half the
                               // same-address atomics per code): hist 12.5 -> 11.2 us, scatter 25.4 -> 19.2 us at 2^20 rows, C = 1024; 1024: the same
#endif
#define VQ_SORT_ROWS_PER_BLOCK (16 * VQ_SORT_THREADS)

// ATen CPU lerp (vectorised form): |w| < 0.5 ? fma(w, end - start, start) : fma(w - 1, end - start, end)
__device__ __forceinline__ float aten_lerp(float start, float end, float w)
{
    const float diff = end - start;
    return (fabsf(w) < 0.5f) ? __builtin_fmaf(w, diff, start) : __builtin_fmaf(w - 1.0f, diff, end);
}

// total = cluster_size.sum() in ATen's cascade order (8 lanes x 4 ILP chains, 4 cascade levels);
// denom[c] = (cs + eps) / (total + C eps) * total        (vqp.py:152-154, 577).  cs: C floats readable by every thread of the
// workgroup (LDS, or global for C > 16384); part [32] and total_s in LDS; NT = threads of the workgroup (all of them call).
template <int NT>
__device__ __forceinline__ void ema_denom_block(const float *cs, int C, float eps, float ceps, float *denom, float *part, float *total_s)
{
    const int tid = threadIdx.x;
    if (tid < 32) {
        const int V = C >> 3;
        const int size = V >> 2;
        int cl2 = 0;
        while ((1 << cl2) < size) cl2++;
        int lp = cl2 / 4;
        if (lp < 4) lp = 4;
        const int step = 1 << lp;
        const int lmask = step - 1;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int i = 0;
        for (; i + step <= size;) {
            for (int jj = 0; jj < step; ++jj, ++i) a0 += cs[32 * i + tid];
            a1 += a0; a0 = 0.f;
            if ((i & (lmask << lp)) == 0) {
                a2 += a1; a1 = 0.f;
                if ((i & (lmask << (2 * lp))) == 0) { a3 += a2; a2 = 0.f; }
            }
        }
        for (; i < size; ++i) a0 += cs[32 * i + tid];
        a0 += a1; a0 += a2; a0 += a3;
        if (tid < 8)
            for (int v = size * 4; v < V; ++v) a0 += cs[v * 8 + tid];
        part[tid] = a0;
    }
    __syncthreads();
    if (tid == 0) {
        float fin = 0.f;
        for (int e = (C >> 3) << 3; e < C; ++e) fin += cs[e];
        for (int l = 0; l < 8; ++l) fin += ((part[l] + part[8 + l]) + part[16 + l]) + part[24 + l];
        *total_s = fin;
    }
    __syncthreads();
    const float total = *total_s;
    const float den = total + ceps;
    for (int c = tid; c < C; c += NT) denom[c] = ((cs[c] + eps) / den) * total;
}

struct SortArgs {
    const int64_t *idx;
    int64_t idx_stride;
    const uint8_t *row_mask;
    int64_t N;
    int C;
    int *hist;     // [C]
    int *cursor;   // [C * 16]
    int *seg_off;  // [C + 1]
    int *chunk_off;  // [C + 1]
    int *perm;     // [N]
    float *count;  // [C] fp32, accumulated into
    int direct;    // 1: one global atomic per row, 256 rows per workgroup (few rows per code: a workgroup's LDS histogram would hold
                   // one or two rows per bin and cost C more atomics to flush); 0: LDS histogram per VQ_SORT_ROWS_PER_BLOCK rows
    // fused train step (vqhip_vq_train_step, no collective between the statistics and the fold): the scan kernel -- the one
    // workgroup that sees every count -- also folds them into cluster_size (ema_inplace, vqp.py:76-97 at :610) and forms the
    // Laplace-smoothed denominators of update_ema (vqp.py:576-584).  cs null: not fused.  C <= 8192 (LDS copy of cluster_size).
    float *cs;     // [C] cluster_size, in place
    float *denom;  // [C] out
    float omd, eps, ceps;
    int fold_next; // the segmented sum that follows folds every code's row itself (SegArgs.fold): every code gets at least one work item
                   // (an empty one for a code without rows), the histogram is handed on zeroed as the codes' tickets, cursor[1] as the
                   // workgroups' ticket
    // several heads in one launch (vqhip_ema_accumulate_batched: blockIdx.y = head): byte strides between consecutive heads' index
    // rows, workspaces (hist / cursor / seg_off / chunk_off / perm share one) and count vectors
    int heads;
    int64_t hs_idx, hs_ws, hs_count;
};

__device__ __forceinline__ SortArgs sort_head_args(const SortArgs &a0)
{
    if (a0.heads <= 1) return a0;
    SortArgs a = a0;
    const int64_t h = blockIdx.y;
    a.idx = (const int64_t *)((const char *)a0.idx + h * a0.hs_idx);
    a.hist = (int *)((char *)a0.hist + h * a0.hs_ws);
    a.cursor = (int *)((char *)a0.cursor + h * a0.hs_ws);
    a.seg_off = (int *)((char *)a0.seg_off + h * a0.hs_ws);
    a.chunk_off = (int *)((char *)a0.chunk_off + h * a0.hs_ws);
    a.perm = (int *)((char *)a0.perm + h * a0.hs_ws);
    a.count = (float *)((char *)a0.count + h * a0.hs_count);
    return a;
}

__device__ __forceinline__ int sort_code(const SortArgs &a, int64_t row)
{
    const int64_t ci = a.idx[row * a.idx_stride];
    const bool ok = (ci >= 0) & (ci < a.C) & (!a.row_mask || a.row_mask[row] != 0);
    return ok ? (int)ci : -1;
}

__global__ void __launch_bounds__(VQ_SORT_THREADS) vq_hist_kernel(const SortArgs a0)
{
    const SortArgs a = sort_head_args(a0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *lh = (int *)smem;
    const bool use_lds = !a.direct;
    const int tid = threadIdx.x, NT = (int)blockDim.x;     // (direct: 256 threads; LDS path: VQ_SORT_THREADS)
    if (use_lds) {
        for (int c = tid; c < a.C; c += NT) lh[c] = 0;
        __syncthreads();
    }
    const int rpb = a.direct ? 256 : VQ_SORT_ROWS_PER_BLOCK;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = min(a.N, r0 + rpb);
    for (int64_t row = r0 + tid; row < r1; row += NT) {
        const int c = sort_code(a, row);
        if (c >= 0) {
            if (use_lds) atomicAdd(&lh[c], 1);
            else atomicAdd(&a.hist[c], 1);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int c = tid; c < a.C; c += NT)
            if (lh[c]) atomicAdd(&a.hist[c], lh[c]);
    }
}

__global__ void __launch_bounds__(1024) vq_scan_kernel(const SortArgs a0)
{
    const SortArgs a = sort_head_args(a0);
    __shared__ int s_cnt[1024];
    __shared__ int s_chk[1024];
    __shared__ float s_part[32];
    __shared__ float s_total;
    extern __shared__ float s_cs[];        // C floats when a.cs is given
    const int tid = threadIdx.x;
    const int per = (a.C + 1023) / 1024;
    const int c_lo = min(a.C, tid * per), c_hi = min(a.C, c_lo + per);
    int sc = 0, sk = 0;
    for (int c = c_lo; c < c_hi; ++c) {
        const int n = a.hist[c];
        sc += n;
        sk += (a.fold_next && n == 0) ? 1 : (n + VQ_SEG_CH - 1) / VQ_SEG_CH;
    }
    s_cnt[tid] = sc;
    s_chk[tid] = sk;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan
        int vc = 0, vk = 0;
        if (tid >= o) { vc = s_cnt[tid - o]; vk = s_chk[tid - o]; }
        __syncthreads();
        s_cnt[tid] += vc;
        s_chk[tid] += vk;
        __syncthreads();
    }
    int oc = s_cnt[tid] - sc, ok = s_chk[tid] - sk;  // exclusive prefix of this thread's range
    for (int c = c_lo; c < c_hi; ++c) {
        const int n = a.hist[c];
        a.seg_off[c] = oc;
        a.chunk_off[c] = ok;
        a.cursor[c * 16] = oc;
        float tot = a.count[c];            // rows of this code seen so far this step (earlier row chunks of a pipelined step; else 0)
        if (n) { tot += (float)n; a.count[c] = tot; }      // integers below 2^24: exact
        if (a.cs) {                        // (the last chunk's scan: `tot` IS the batch's count)
            const float v = aten_lerp(a.cs[c], tot, a.omd);
            a.cs[c] = v;
            s_cs[c] = v;
        }
        oc += n;
        ok += (a.fold_next && n == 0) ? 1 : (n + VQ_SEG_CH - 1) / VQ_SEG_CH;
        if (a.fold_next) a.hist[c] = 0;
    }
    if (tid == 1023) {
        a.seg_off[a.C] = s_cnt[1023];
        a.chunk_off[a.C] = s_chk[1023];
    }
    if (a.fold_next && tid <= 32 && tid <= a.C) {      // the workgroups' tickets: cursor[1], and cursor[16 g + 2] for the groups g < min(C, 32)
        if (tid == 0) a.cursor[1] = 0;
        else a.cursor[16 * (tid - 1) + 2] = 0;
    }
    if (a.cs) {
        __syncthreads();
        ema_denom_block<1024>(s_cs, a.C, a.eps, a.ceps, a.denom, s_part, &s_total);
    }
}

__global__ void __launch_bounds__(VQ_SORT_THREADS) vq_scatter_kernel(const SortArgs a0)
{
    const SortArgs a = sort_head_args(a0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *lc = (int *)smem;
    const bool use_lds = !a.direct;
    const int tid = threadIdx.x;
    const int rpb = a.direct ? 256 : VQ_SORT_ROWS_PER_BLOCK;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int64_t r1 = min(a.N, r0 + rpb);
    constexpr int RPT = 16, NT = VQ_SORT_THREADS;
    if (!use_lds) {
        for (int64_t row = r0 + tid; row < r1; row += 256) {
            const int c = sort_code(a, row);
            if (c >= 0) a.perm[atomicAdd(&a.cursor[c * 16], 1)] = (int)row;
        }
        return;
    }
    for (int c = tid; c < a.C; c += NT) lc[c] = 0;
    __syncthreads();
    int code[RPT], rank[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int64_t row = r0 + tid + NT * u;
        code[u] = (row < r1) ? sort_code(a, row) : -1;
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) rank[u] = (code[u] >= 0) ? atomicAdd(&lc[code[u]], 1) : 0;
    __syncthreads();
    for (int c = tid; c < a.C; c += NT) {
        const int n = lc[c];
        if (n) lc[c] = atomicAdd(&a.cursor[c * 16], n);  // reserve n slots; lc[c] becomes the base
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RPT; ++u)
        if (code[u] >= 0) a.perm[lc[code[u]] + rank[u]] = (int)(r0 + tid + NT * u);
}

struct SegArgs {
    const void *x;
    int D;
    int64_t ldx;
    const float *rnorm;
    int cosine;
    int C;
    const int *seg_off;
    const int *chunk_off;
    const int *perm;
    float *embed_sum;
    const void *qsrc;          // nullable: codebook rows in x's dtype, [C, D] -- the q of the commitment loss
    double *sqerr_partial;     // nullable: one entry per work item (n_partial of them)
    int64_t n_partial;
    // FOLD (fused train step, one head): the wave that adds the LAST chunk of a code folds the code's sum into embed_avg and renormalises
    // embed (ema_embed_row: what vq_step_fold_kernel did in a launch of its own); the workgroup that finishes last reduces the loss'
    // partials [0, n_loss) (vq_reduce_kernel's order).  code_done [C] and wg_done: zeroed tickets (vq_scan_kernel, SortArgs.fold_next).
    int *code_done, *wg_done;
    float *embed_avg, *embed;
    const float *denom;
    float omd;
    int fold_cosine;
    const double *loss_partials;
    int64_t n_loss;
    double loss_scale;
    float *loss_out;
    // several heads in one launch (blockIdx.y = head): byte strides of the rows, the workspace, embed_sum, the loss' code rows, the partials
    int heads;
    int64_t hs_x, hs_ws, hs_sum, hs_qsrc, hs_sq;
};

__device__ __forceinline__ SegArgs seg_head_args(const SegArgs &a0)
{
    if (a0.heads <= 1) return a0;
    SegArgs a = a0;
    const int64_t h = blockIdx.y;
    a.x = (const char *)a0.x + h * a0.hs_x;
    a.seg_off = (const int *)((const char *)a0.seg_off + h * a0.hs_ws);
    a.chunk_off = (const int *)((const char *)a0.chunk_off + h * a0.hs_ws);
    a.perm = (const int *)((const char *)a0.perm + h * a0.hs_ws);
    a.embed_sum = (float *)((char *)a0.embed_sum + h * a0.hs_sum);
    if (a0.qsrc) a.qsrc = (const char *)a0.qsrc + h * a0.hs_qsrc;
    if (a0.sqerr_partial) a.sqerr_partial = (double *)((char *)a0.sqerr_partial + h * a0.hs_sq);
    return a;
}

// EPL = elements per lane (vector path: D == 64 * EPL * nvec ... handled by the h loop)
template <bool XBF16, bool VEC>
__global__ void __launch_bounds__(256) vq_segsum_kernel(const SegArgs a0)
{
    const SegArgs a = seg_head_args(a0);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);   // work item = (code, chunk)
    const int total = a.chunk_off[a.C];
    if (w >= total) return;
    // binary search: largest c with chunk_off[c] <= w   (chunk_off is non-decreasing)
    int lo = 0, hi = a.C;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.chunk_off[mid] <= w) lo = mid; else hi = mid;
    }
    const int c = __builtin_amdgcn_readfirstlane(lo);
    const int j = w - a.chunk_off[c];
    const int beg = a.seg_off[c] + j * VQ_SEG_CH;
    const int end = min(a.seg_off[c + 1], beg + VQ_SEG_CH);

    constexpr int NH = 2;          // up to 512 columns: 2 x (64 lanes x 4)
    float acc[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[h][e] = 0.f;

    constexpr int U = 8;
    for (int r = beg; r < end; r += U) {
        int rows[U];
#pragma unroll
        for (int u = 0; u < U; ++u) rows[u] = a.perm[min(r + u, end - 1)];   // wave-uniform
        float v[U][NH][4];
        float rn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rn[u] = a.cosine ? a.rnorm[rows[u]] : 1.f;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int d = h * 256 + lane * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][h][e] = 0.f;
                if (VEC) {
                    if (d < a.D) {
                        if (XBF16) {
                            const uint2 wv = *(const uint2 *)((const unsigned short *)a.x + (int64_t)rows[u] * a.ldx + d);
                            v[u][h][0] = __uint_as_float(wv.x << 16); v[u][h][1] = __uint_as_float(wv.x & 0xffff0000u);
                            v[u][h][2] = __uint_as_float(wv.y << 16); v[u][h][3] = __uint_as_float(wv.y & 0xffff0000u);
                        } else {
                            const f32x4 wv = *(const f32x4 *)((const float *)a.x + (int64_t)rows[u] * a.ldx + d);
                            v[u][h][0] = wv.x; v[u][h][1] = wv.y; v[u][h][2] = wv.z; v[u][h][3] = wv.w;
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (d + e < a.D) v[u][h][e] = load_elem<XBF16>(a.x, (int64_t)rows[u] * a.ldx + d + e);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u < end) {
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[u][h][e];
                        if (a.cosine) {
                            t = t / rn[u];
                            if (XBF16) t = round_to_bf16(t);
                        }
                        acc[h][e] += t;
                    }
            }
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int d = h * 256 + lane * 4 + e;
            if (d < a.D && acc[h][e] != 0.f) unsafeAtomicAdd(&a.embed_sum[(size_t)c * a.D + d], acc[h][e]);
        }
}

template <bool COHERENT>
__device__ __forceinline__ void ema_embed_row(float *embed_avg, float *embed, const float *embed_sum,
                                              const float *weight, const float *denom, int C, int D,
                                              float omd, int cosine, int do_lerp, int do_update, int c);

// Common case (D <= 256, vector-aligned rows, no per-row normalisation): the rows stay packed in their load registers
// (2 VGPRs per bf16 row, 4 per fp32 row), so 16 rows are in flight per wave instead of 8.
// SQ: this pass reads every (unmasked) row next to its code, which is all the commitment loss needs (F.mse_loss(quantize, x),
// vqp.py:1327): the wave also sums ||q_c - x||^2 over its rows -- an fp32 FMA chain over the batch's elements of a lane, then in
// double per batch of rows in flight (relative error of the total ~1e-8; the loss is held to 1e-5) -- so the search does not have to
// re-read x for it.
template <bool XBF16, bool SQ, bool FOLD>
__global__ void __launch_bounds__(256) vq_segsum_fast_kernel(const SegArgs a0)
{
    const SegArgs a = seg_head_args(a0);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int total = a.chunk_off[a.C];
    // (FOLD: the partials are read by another workgroup of this launch: device-scope stores, straight to the memory side)
    auto put_partial = [&](double v) {
        if (FOLD) __hip_atomic_store(&a.sqerr_partial[w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a.sqerr_partial[w] = v;
    };
    if (w >= total) {
        if (SQ && lane == 0 && w < a.n_partial) put_partial(0.0);
        if (!FOLD) return;
    }
    if (!FOLD || w < total) {
    int lo = 0, hi = a.C;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.chunk_off[mid] <= w) lo = mid; else hi = mid;
    }
    const int c = __builtin_amdgcn_readfirstlane(lo);
    const int j = w - a.chunk_off[c];
    const int beg = a.seg_off[c] + j * VQ_SEG_CH;
    const int end = min(a.seg_off[c + 1], beg + VQ_SEG_CH);
#ifndef VQ_SEG_U_BF
#define VQ_SEG_U_BF 16
#endif
#ifndef VQ_SEG_U_F32
#define VQ_SEG_U_F32 16
#endif
    constexpr int U = XBF16 ? VQ_SEG_U_BF : VQ_SEG_U_F32;   // rows in flight per wave
    double sq = 0.0;
    for (int d0 = 0; d0 < a.D; d0 += 256) {                 // D <= 512: one or two 256-element column blocks
        const int d = d0 + lane * 4;
        const bool act = d < a.D;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float g[4] = {0.f, 0.f, 0.f, 0.f};                  // this lane's 4 elements of the code's row (the loss' q)
        if (SQ && act) {
            if (XBF16) {
                const uint2 gw = *(const uint2 *)((const unsigned short *)a.qsrc + (size_t)c * a.D + d);
                g[0] = __uint_as_float(gw.x << 16); g[1] = __uint_as_float(gw.x & 0xffff0000u);
                g[2] = __uint_as_float(gw.y << 16); g[3] = __uint_as_float(gw.y & 0xffff0000u);
            } else {
                const f32x4 gw = *(const f32x4 *)((const float *)a.qsrc + (size_t)c * a.D + d);
                g[0] = gw.x; g[1] = gw.y; g[2] = gw.z; g[3] = gw.w;
            }
        }
        for (int r = beg; r < end; r += U) {
            int rows[U];
#pragma unroll
            for (int u = 0; u < U; ++u) rows[u] = a.perm[min(r + u, end - 1)];   // wave-uniform
            if (XBF16) {
                uint2 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (act) v[u] = *(const uint2 *)((const unsigned short *)a.x + (int64_t)rows[u] * a.ldx + d);
                float bsq = 0.f;            // this batch's squared error in fp32 (64 terms), folded into the double once per batch
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (act && r + u < end) {
                        const float x0 = __uint_as_float(v[u].x << 16), x1 = __uint_as_float(v[u].x & 0xffff0000u);
                        const float x2 = __uint_as_float(v[u].y << 16), x3 = __uint_as_float(v[u].y & 0xffff0000u);
                        acc[0] += x0; acc[1] += x1; acc[2] += x2; acc[3] += x3;
                        if (SQ) {
                            const float e0 = g[0] - x0, e1 = g[1] - x1, e2 = g[2] - x2, e3 = g[3] - x3;
                            bsq = __builtin_fmaf(e0, e0, bsq); bsq = __builtin_fmaf(e1, e1, bsq);
                            bsq = __builtin_fmaf(e2, e2, bsq); bsq = __builtin_fmaf(e3, e3, bsq);
                        }
                    }
                if (SQ) sq += (double)bsq;
            } else {
                f32x4 v[U];
                // loads issued back to back from always-valid addresses (an inactive lane re-reads the row's first elements; the
                // rows past the chunk's end repeat its last row), consumed through selects: no branch per row, so the compiler can
                // wait for the rows one by one (vmcnt(U - 1 - u)) while the later ones are still in flight -- 272 -> 247 us for the
                // statistics of 2^20 fp32 rows; the same form for bf16 rows measured no gain (161 vs 160-167 us) and is not used
                const int dd = act ? d : 0;
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = *(const f32x4 *)((const float *)a.x + (int64_t)rows[u] * a.ldx + dd);
                float bsq = 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool on = act && r + u < end;
                    const float x0 = on ? v[u].x : 0.f, x1 = on ? v[u].y : 0.f, x2 = on ? v[u].z : 0.f, x3 = on ? v[u].w : 0.f;
                    acc[0] += x0; acc[1] += x1; acc[2] += x2; acc[3] += x3;
                    if (SQ) {
                        const float e0 = on ? g[0] - x0 : 0.f, e1 = on ? g[1] - x1 : 0.f, e2 = on ? g[2] - x2 : 0.f, e3 = on ? g[3] - x3 : 0.f;
                        bsq = __builtin_fmaf(e0, e0, bsq); bsq = __builtin_fmaf(e1, e1, bsq);
                        bsq = __builtin_fmaf(e2, e2, bsq); bsq = __builtin_fmaf(e3, e3, bsq);
                    }
                }
                if (SQ) sq += (double)bsq;
            }
        }
        if (act)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (acc[e] != 0.f) unsafeAtomicAdd(&a.embed_sum[(size_t)c * a.D + d + e], acc[e]);
    }
    if (SQ) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sq += __shfl_xor(sq, o, 64);
        if (lane == 0) put_partial(sq);
    }
    if (FOLD) {
        // This wave's adds have been performed (fp32 atomics and device-scope stores execute at the memory side, past the XCD's L2, and
        // are acknowledged from there: vmcnt) before its ticket is drawn; the wave that draws a code's LAST ticket reads the sums with
        // device-scope loads.  Deliberately NOT __threadfence(): on this chip that is a write-back + invalidate of the XCD's whole L2 per
        // call -- measured: 127 -> 530 us for this kernel with one fence per wave.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last = 0;
        if (lane == 0) last = atomicAdd(&a.code_done[c], 1) == (a.chunk_off[c + 1] - a.chunk_off[c]) - 1;
        if (__builtin_amdgcn_readfirstlane(last))
            ema_embed_row<true>(a.embed_avg, a.embed, a.embed_sum, nullptr, a.denom, a.C, a.D, a.omd, a.fold_cosine, 1, 1, c);
    }
    }
    if (FOLD) {
        // the workgroup that finishes last (every partial of the launch, and of the earlier row chunks' launches, is in memory by then)
        // reduces the commitment loss: fp64, the fixed order of vq_reduce_kernel
        // (two ticket levels: the workgroups of a launch finish together, and same-address atomics retire one per ~45 cycles -- 1 280
        // tickets on one counter would be a 27 us tail; 32 counters of <= 40 and one of 32 are not)
        __shared__ double red[256];
        __shared__ int is_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int G = a.C < 32 ? a.C : 32, nb = (int)gridDim.x, grp = (int)blockIdx.x % G;
            const int members = nb / G + (grp < nb % G ? 1 : 0);
            int l = 0;
            if (atomicAdd(a.wg_done + 16 * grp + 1, 1) == members - 1) l = atomicAdd(a.wg_done, 1) == (nb < G ? nb : G) - 1;
            is_last = l;
        }
        __syncthreads();
        if (!is_last || !a.loss_out) return;
        double sum = 0.0;
        for (int64_t i = threadIdx.x; i < a.n_loss; i += 256)
            sum += __hip_atomic_load(&a.loss_partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[threadIdx.x] = sum;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) *a.loss_out = (float)(red[0] * a.loss_scale);
    }
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t vqhip_ema_workspace_bytes(int64_t N, int C)
{
    if (N < 0 || C <= 0) return 0;
    // hist[C] | cursor[16 C] | seg_off[C+1] | chunk_off[C+1] | perm[N]   (all int32, 256-byte aligned pieces)
    return align_up((size_t)C * 4, 256) + align_up((size_t)C * 64, 256) + 2 * align_up(((size_t)C + 1) * 4, 256) +
           align_up((size_t)(N > 0 ? N : 1) * 4, 256);
}

static inline int64_t seg_work_items(int64_t N, int C)
{
    const int64_t max_items = N / VQ_SEG_CH + C + 1;       // sum_c ceil(n_c / CH) <= N / CH + C
    return (max_items + 3) / 4 * 4;                        // whole workgroups of 4 waves
}

extern "C" int64_t vqhip_ema_sqerr_partials(int64_t N, int C) { return (N < 0 || C <= 0) ? 0 : seg_work_items(N, C); }

// what the fused train step adds to the statistics pass: hist_zeroed -- the workspace's histogram (its first C ints) was zeroed by
// the caller on this stream, no memset launch; cs / denom -- the scan kernel folds the counts into cluster_size and forms
// update_ema's denominators (SortArgs)
struct StatsFuse {
    int hist_zeroed;
    float *cs, *denom;
    float omd, eps;
    // several heads (vqhip_ema_accumulate_batched): byte strides; count / embed_sum / workspace / qsrc / sqerr_partial of head h sit h
    // strides behind head 0's
    int heads = 1;
    int64_t hs_x = 0, hs_idx = 0, hs_ws = 0, hs_stats = 0, hs_qsrc = 0, hs_sq = 0;
    // fold of embed_avg / embed and the loss' reduction inside the segmented sum (SegArgs.fold; needs cs / denom above)
    float *fold_embed_avg = nullptr, *fold_embed = nullptr;
    int fold_cosine = 0;
    const double *loss_partials = nullptr;
    int64_t n_loss = 0;
    double loss_scale = 0.0;
    float *loss_out = nullptr;
};

static int ema_accumulate_impl(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                               const int64_t *idx, int64_t idx_stride, const float *rnorm, int metric,
                               const uint8_t *row_mask, int C,
                               float *count, float *embed_sum, void *workspace, size_t workspace_bytes,
                               const void *qsrc, double *sqerr_partial, void *stream, const StatsFuse *fuse = nullptr)
{
    if (N < 0 || C <= 0 || D < 1) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: bad size");
    if (N == 0) {
        if (sqerr_partial) {
            hipError_t e0 = hipMemsetAsync(sqerr_partial, 0, (size_t)seg_work_items(0, C) * sizeof(double), (hipStream_t)stream);
            if (e0 != hipSuccess) VQ_FAIL((int)e0, "hipMemsetAsync(sqerr_partial): %s", hipGetErrorString(e0));
        }
        return 0;
    }
    if (!x || !idx || !count || !embed_sum || !workspace) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: null pointer");
    if (D > VQ_WIDE_MAX_D) VQ_FAIL(VQHIP_EDIM, "ema_accumulate: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (N >= ((int64_t)1 << 31)) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: N must be < 2^31");
    if (metric == VQHIP_COSINE && !rnorm) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: cosine needs rnorm");
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: unknown dtype");
    if (workspace_bytes < vqhip_ema_workspace_bytes(N, C)) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: workspace too small");
    if (((uintptr_t)workspace) & 255) VQ_FAIL(VQHIP_EALIGN, "ema_accumulate: workspace must be 256-byte aligned");

    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    SortArgs s;
    s.idx = idx; s.idx_stride = idx_stride; s.row_mask = row_mask; s.N = N; s.C = C; s.count = count;
    s.hist = (int *)ws;           ws += align_up((size_t)C * 4, 256);
    s.cursor = (int *)ws;         ws += align_up((size_t)C * 64, 256);
    s.seg_off = (int *)ws;        ws += align_up(((size_t)C + 1) * 4, 256);
    s.chunk_off = (int *)ws;      ws += align_up(((size_t)C + 1) * 4, 256);
    s.perm = (int *)ws;

    const bool hist_zeroed = fuse && fuse->hist_zeroed;
    s.cs = fuse ? fuse->cs : nullptr;
    s.denom = fuse ? fuse->denom : nullptr;
    s.omd = fuse ? fuse->omd : 0.f;
    s.eps = fuse ? fuse->eps : 0.f;
    s.ceps = fuse ? (float)((double)C * (double)fuse->eps) : 0.f;
    const bool fold = fuse && fuse->fold_embed_avg && fuse->fold_embed && s.cs && s.denom;
    s.fold_next = fold ? 1 : 0;
    if (s.cs && (C > 8192 || !s.denom)) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: the fused cluster-size fold needs C <= 8192 and a denominator buffer");
    const unsigned nh = (unsigned)((fuse && fuse->heads > 1) ? fuse->heads : 1);
    s.heads = (int)nh;
    s.hs_idx = fuse ? fuse->hs_idx : 0; s.hs_ws = fuse ? fuse->hs_ws : 0; s.hs_count = fuse ? fuse->hs_stats : 0;
    if (nh > 1 && (rnorm || metric == VQHIP_COSINE)) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: batched heads take unit-norm or Euclidean rows");
    if (!hist_zeroed) {
        hipError_t e = nh > 1 ? hipMemset2DAsync(s.hist, (size_t)s.hs_ws, 0, (size_t)C * 4, nh, st) : hipMemsetAsync(s.hist, 0, (size_t)C * 4, st);
        if (e != hipSuccess) VQ_FAIL((int)e, "hipMemsetAsync(hist): %s", hipGetErrorString(e));
    }
    // LDS histograms, one global atomic per (workgroup, code), while the histogram fits.  (Round 3 tried the direct path -- one
    // returning global atomic per row from a full grid -- for few rows per code, N < 128 C: cfg 5 15.5 -> 19.7 ms, cfg 4 shard
    // 3.02 -> 3.15 ms.  Per-row returning atomics are slower than the flush of a sparse LDS histogram.)
    s.direct = (C > VQ_HIST_LDS_MAX) ? 1 : 0;
    const int rpb = s.direct ? 256 : VQ_SORT_ROWS_PER_BLOCK;
    const unsigned sort_blocks = (unsigned)((N + rpb - 1) / rpb);
    const int lds = s.direct ? 0 : C * 4;
    const unsigned sort_threads = s.direct ? 256u : (unsigned)VQ_SORT_THREADS;
    hipLaunchKernelGGL(vq_hist_kernel, dim3(sort_blocks, nh), dim3(sort_threads), lds, st, s);
    hipLaunchKernelGGL(vq_scan_kernel, dim3(1, nh), dim3(1024), s.cs ? (size_t)C * 4 : 0, st, s);
    hipLaunchKernelGGL(vq_scatter_kernel, dim3(sort_blocks, nh), dim3(sort_threads), lds, st, s);

    SegArgs g;
    g.x = x; g.D = D; g.ldx = ldx; g.rnorm = rnorm; g.cosine = (metric == VQHIP_COSINE); g.C = C;
    g.seg_off = s.seg_off; g.chunk_off = s.chunk_off; g.perm = s.perm; g.embed_sum = embed_sum;
    g.qsrc = qsrc; g.sqerr_partial = sqerr_partial; g.n_partial = seg_work_items(N, C);
    g.heads = (int)nh;
    g.hs_x = fuse ? fuse->hs_x : 0; g.hs_ws = fuse ? fuse->hs_ws : 0; g.hs_sum = fuse ? fuse->hs_stats : 0;
    g.hs_qsrc = fuse ? fuse->hs_qsrc : 0; g.hs_sq = fuse ? fuse->hs_sq : 0;
    g.code_done = s.hist; g.wg_done = s.cursor + 1;
    g.embed_avg = fold ? fuse->fold_embed_avg : nullptr; g.embed = fold ? fuse->fold_embed : nullptr;
    g.denom = s.denom; g.omd = s.omd; g.fold_cosine = fold ? fuse->fold_cosine : 0;
    g.loss_partials = fold ? fuse->loss_partials : nullptr; g.n_loss = fold ? fuse->n_loss : 0;
    g.loss_scale = fold ? fuse->loss_scale : 0.0; g.loss_out = fold ? fuse->loss_out : nullptr;
    if (nh > 1 && s.cs) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: the fused cluster-size fold is for one head");
    const unsigned seg_blocks = (unsigned)(seg_work_items(N, C) / 4);
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    const bool vec = (D % 4 == 0) && (((uintptr_t)x) % (4 * es) == 0) && ((ldx * es) % (4 * es) == 0);
    const bool bf = (x_dtype == VQHIP_BF16);
    if (sqerr_partial && !(vec && D <= VQ_WIDE_MAX_D && !g.cosine && qsrc && ((uintptr_t)qsrc) % (4 * es) == 0))
        VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_sqerr: needs the Euclidean metric, D %% 4 == 0, D <= 2048 and 16-byte aligned rows");
    if (D > 512 && !(vec && !g.cosine))
        VQ_FAIL(VQHIP_EDIM, "ema_accumulate: D=%d > 512 needs D %% 4 == 0, rows aligned to 4 elements and Euclidean / unit-norm rows", D);
#ifndef VQ_SEG_SLOW
    if (vec && D <= VQ_WIDE_MAX_D && !g.cosine) {
        if (fold) {
            if (!sqerr_partial || nh > 1 || D > 512) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate: the fused fold needs the loss partials, one head and D <= 512");
            if (bf) hipLaunchKernelGGL((vq_segsum_fast_kernel<true, true, true>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
            else hipLaunchKernelGGL((vq_segsum_fast_kernel<false, true, true>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
        } else if (sqerr_partial) {
            if (bf) hipLaunchKernelGGL((vq_segsum_fast_kernel<true, true, false>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
            else hipLaunchKernelGGL((vq_segsum_fast_kernel<false, true, false>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
        } else {
            if (bf) hipLaunchKernelGGL((vq_segsum_fast_kernel<true, false, false>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
            else hipLaunchKernelGGL((vq_segsum_fast_kernel<false, false, false>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
        }
        return launch_status("vq_ema_accumulate");
    }
#endif
    if (sqerr_partial || fold) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_sqerr: the loss and the fused fold need the fast segment-sum kernel's conditions");
    if (bf && vec) hipLaunchKernelGGL((vq_segsum_kernel<true, true>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
    else if (bf) hipLaunchKernelGGL((vq_segsum_kernel<true, false>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
    else if (vec) hipLaunchKernelGGL((vq_segsum_kernel<false, true>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((vq_segsum_kernel<false, false>), dim3(seg_blocks, nh), dim3(256), 0, st, g);
    return launch_status("vq_ema_accumulate");
}

extern "C" int vqhip_ema_accumulate(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                                    const int64_t *idx, int64_t idx_stride, const float *rnorm, int metric,
                                    const uint8_t *row_mask, int C,
                                    float *count, float *embed_sum, void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    return ema_accumulate_impl(x, x_dtype, N, D, ldx, idx, idx_stride, rnorm, metric, row_mask, C, count, embed_sum, workspace,
                               workspace_bytes, nullptr, nullptr, stream);
}

extern "C" int vqhip_ema_accumulate_sqerr(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                                          const int64_t *idx, int64_t idx_stride, const uint8_t *row_mask, int C,
                                          float *count, float *embed_sum, void *workspace, size_t workspace_bytes,
                                          const float *packed, const float *embed, double *sqerr_partial, void *stream)
{
    if (!sqerr_partial || !packed || !embed) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_sqerr: null pointer");
    if (D < 1 || D > VQ_WIDE_MAX_D || C <= 0) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_sqerr: bad size");
    // the loss' q rows: what the search writes for this dtype -- fp32 rows: embed; bf16 rows: the packed codebook's bf16 copy
    const void *qsrc = (x_dtype == VQHIP_BF16) ? (const void *)((const char *)packed + vq_packed_bf16_offset(C, D)) : (const void *)embed;
    return ema_accumulate_impl(x, x_dtype, N, D, ldx, idx, idx_stride, nullptr, VQHIP_EUCLID, row_mask, C, count, embed_sum, workspace,
                               workspace_bytes, qsrc, sqerr_partial, stream);
}

// The same pass for a caller that has zeroed the histogram -- the first C ints of `workspace` -- itself: a residual VQ zeroes the
// workspaces of all its stages in one launch before the loop, instead of one memset per stage queued on the statistics stream
// behind a chip-filling search.  packed / embed / sqerr_partial may be null together (statistics only).
extern "C" int vqhip_ema_accumulate_prezeroed(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                                              const int64_t *idx, int64_t idx_stride, const uint8_t *row_mask, int C,
                                              float *count, float *embed_sum, void *workspace, size_t workspace_bytes,
                                              const float *packed, const float *embed, double *sqerr_partial, void *stream)
{
    if (D < 1 || D > VQ_WIDE_MAX_D || C <= 0) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_prezeroed: bad size");
    if (sqerr_partial && (!packed || !embed)) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_prezeroed: the loss needs packed and embed");
    const void *qsrc = !sqerr_partial ? nullptr
                     : (x_dtype == VQHIP_BF16) ? (const void *)((const char *)packed + vq_packed_bf16_offset(C, D)) : (const void *)embed;
    StatsFuse f;
    f.hist_zeroed = 1; f.cs = nullptr; f.denom = nullptr; f.omd = 0.f; f.eps = 0.f;
    return ema_accumulate_impl(x, x_dtype, N, D, ldx, idx, idx_stride, nullptr, VQHIP_EUCLID, row_mask, C, count, embed_sum, workspace,
                               workspace_bytes, qsrc, sqerr_partial, stream, &f);
}

// The statistics of H heads in one set of launches (blockIdx.y = head): x [H, N, D] at x_hstride elements between heads, idx [H, N],
// stats [H, stats_stride] = embed_sum [C, D] || count [C] per head (ACCUMULATED INTO: zero it first), workspace H x
// vqhip_ema_batched_ws_stride(N, C) bytes, packed / embed (nullable together with sqerr_partial [H, vqhip_ema_sqerr_partials(N, C)]) the
// codebooks as vqhip_pack_codebook_batched laid them out / [H, C, D].  Euclidean, or cosine on unit-norm rows.
extern "C" size_t vqhip_ema_batched_ws_stride(int64_t N, int C) { return align_up(vqhip_ema_workspace_bytes(N, C), 256); }

extern "C" int vqhip_ema_accumulate_batched(const void *x, int x_dtype, int H, int64_t N, int D, int64_t ldx, int64_t x_hstride,
                                            const int64_t *idx, const uint8_t *row_mask, int C, float *stats, int64_t stats_stride,
                                            void *workspace, size_t workspace_bytes, const float *packed, const float *embed,
                                            double *sqerr_partial, void *stream)
{
    if (H < 1 || D < 1 || D > 512 || C <= 0) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_batched: bad size");
    if (!stats || stats_stride < (int64_t)C * D + C) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_batched: stats null or stats_stride smaller than C D + C");
    if (sqerr_partial && (!packed || !embed)) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_batched: the loss needs packed and embed");
    const size_t wss = vqhip_ema_batched_ws_stride(N, C);
    if (workspace_bytes < wss * (size_t)H) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_batched: workspace too small");
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    StatsFuse f;
    f.hist_zeroed = 0; f.cs = nullptr; f.denom = nullptr; f.omd = 0.f; f.eps = 0.f;
    f.heads = H;
    f.hs_x = x_hstride * es; f.hs_idx = N * 8; f.hs_ws = (int64_t)wss; f.hs_stats = stats_stride * 4;
    f.hs_qsrc = (x_dtype == VQHIP_BF16) ? (int64_t)vq_packed_total_bytes(C, D) : (int64_t)C * D * 4;
    f.hs_sq = seg_work_items(N, C) * (int64_t)sizeof(double);
    const void *qsrc = !sqerr_partial ? nullptr
                     : (x_dtype == VQHIP_BF16) ? (const void *)((const char *)packed + vq_packed_bf16_offset(C, D)) : (const void *)embed;
    return ema_accumulate_impl(x, x_dtype, N, D, ldx, idx, 1, nullptr, VQHIP_EUCLID, row_mask, C, stats + (size_t)C * D, stats, workspace,
                               wss, qsrc, sqerr_partial, stream, &f);
}

// The statistics of S consecutive stages of a residual VQ in ONE set of launches (blockIdx.y = stage): stage s reads its input rows at
// x + s * x_sstride elements (the stage inputs a residual chain materialised one behind the other) and its codes in column s of
// idx [N, idx_stride].  Why: issued per stage on a side stream these passes are 4 short launches each, and beside a screening kernel
// that fills every SIMD's register file (two 256-register waves) each of them waits for a workgroup slot while slowing the search
// down (profiles/r5_rvq_cfg3); batched they are four full-size launches behind the loop.
extern "C" int vqhip_ema_accumulate_stages(const void *x, int x_dtype, int S, int64_t N, int D, int64_t ldx, int64_t x_sstride,
                                           const int64_t *idx, int64_t idx_stride, const uint8_t *row_mask, int C, float *stats,
                                           int64_t stats_stride, void *workspace, size_t workspace_bytes, int hist_zeroed,
                                           const float *packed, int64_t packed_sstride, const float *embed, int64_t embed_sstride,
                                           double *sqerr_partial, int64_t sqerr_stride, void *stream)
{
    if (S < 1 || D < 1 || D > 512 || C <= 0) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_stages: bad size");
    if (!stats || stats_stride < (int64_t)C * D + C) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_stages: stats null or stats_stride smaller than C D + C");
    if (idx_stride < S) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_stages: idx_stride smaller than the number of stages");
    if (sqerr_partial && (!packed || !embed || sqerr_stride < seg_work_items(N, C))) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_stages: the loss needs packed, embed and sqerr_stride >= vqhip_ema_sqerr_partials(N, C)");
    const size_t wss = vqhip_ema_batched_ws_stride(N, C);
    if (workspace_bytes < wss * (size_t)S) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_stages: workspace too small");
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;
    if ((x_sstride * es) & 15) VQ_FAIL(VQHIP_EALIGN, "ema_accumulate_stages: the stages' rows must stay 16-byte aligned");
    StatsFuse f;
    f.hist_zeroed = hist_zeroed ? 1 : 0; f.cs = nullptr; f.denom = nullptr; f.omd = 0.f; f.eps = 0.f;
    f.heads = S;
    f.hs_x = x_sstride * es; f.hs_idx = 8; f.hs_ws = (int64_t)wss; f.hs_stats = stats_stride * 4;
    f.hs_qsrc = (x_dtype == VQHIP_BF16) ? packed_sstride * 4 : embed_sstride * 4;
    f.hs_sq = sqerr_stride * (int64_t)sizeof(double);
    const void *qsrc = !sqerr_partial ? nullptr
                     : (x_dtype == VQHIP_BF16) ? (const void *)((const char *)packed + vq_packed_bf16_offset(C, D)) : (const void *)embed;
    return ema_accumulate_impl(x, x_dtype, N, D, ldx, idx, idx_stride, nullptr, VQHIP_EUCLID, row_mask, C, stats + (size_t)C * D, stats, workspace,
                               wss, qsrc, sqerr_partial, stream, &f);
}

// The statistics of H independent (rows, codebook) pairs in one set of launches with every stride given by the caller (bytes): the G
// groups of a grouped residual VQ at one stage (vq_rvq_chain.hip; rvq.py:634-724 runs them one after the other).  count / embed_sum of
// head h sit h * hs_stats behind head 0's; hist_zeroed as in vqhip_ema_accumulate_prezeroed.  Euclidean.
int vq_ema_accumulate_heads(const void *x, int x_dtype, int H, int64_t N, int D, int64_t ldx, int64_t hs_x, const int64_t *idx,
                            int64_t idx_stride, int64_t hs_idx, const uint8_t *row_mask, int C, float *count, float *embed_sum,
                            int64_t hs_stats, void *workspace, int64_t hs_ws, int hist_zeroed, const void *qsrc, int64_t hs_qsrc,
                            double *sqerr_partial, int64_t hs_sq, void *stream)
{
    if (H < 1 || D < 1 || D > 512 || C <= 0) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_heads: bad size");
    if (hs_ws < (int64_t)vqhip_ema_batched_ws_stride(N, C)) VQ_FAIL(VQHIP_EINVAL, "ema_accumulate_heads: workspace slices too small");
    StatsFuse f;
    f.hist_zeroed = hist_zeroed ? 1 : 0; f.cs = nullptr; f.denom = nullptr; f.omd = 0.f; f.eps = 0.f;
    f.heads = H;
    f.hs_x = hs_x; f.hs_idx = hs_idx; f.hs_ws = hs_ws; f.hs_stats = hs_stats; f.hs_qsrc = hs_qsrc; f.hs_sq = hs_sq;
    return ema_accumulate_impl(x, x_dtype, N, D, ldx, idx, idx_stride, nullptr, VQHIP_EUCLID, row_mask, C, count, embed_sum, workspace,
                               (size_t)hs_ws, qsrc, sqerr_partial, stream, &f);
}

// ------------------------------------------------------------------------------------------------
// EMA fold + codebook renormalisation
// ------------------------------------------------------------------------------------------------
// (hs_*: element strides between consecutive heads of a batched launch, blockIdx.y = head; 0 for a plain one)
__global__ void __launch_bounds__(256) vq_ema_cs_lerp_kernel(float *cs, const float *count, const float *weight, int C, float omd,
                                                             int64_t hs_cs, int64_t hs_count)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    cs += blockIdx.y * hs_cs;
    count += blockIdx.y * hs_count;
    const float w = weight ? omd * weight[c] : omd;
    cs[c] = aten_lerp(cs[c], count[c], w);
}

// (the arithmetic: ema_denom_block above)
__global__ void __launch_bounds__(256) vq_ema_denom_kernel(const float *cs_g, int C, float eps, float ceps, float *denom)
{
    cs_g += (size_t)blockIdx.y * C;        // (batched heads: cluster_size [H, C], denom [H, C])
    denom += (size_t)blockIdx.y * C;
    __shared__ float part[32];
    __shared__ float total_s;
    extern __shared__ float cs_lds[];      // C floats when the launch provides them (C <= 16384), else unused
    const int tid = threadIdx.x;
    // the 32 summation chains are latency-bound on dependent global loads (33 us at C = 4096): stage cluster_size in LDS first
    const float *cs = cs_g;
    if (C <= 16384) {
        for (int c = tid; c < C; c += 256) cs_lds[c] = cs_g[c];
        __syncthreads();
        cs = cs_lds;
    }
    ema_denom_block<256>(cs, C, eps, ceps, denom, part, &total_s);
}

// one wave per code row.  COHERENT: embed_sum was accumulated by other workgroups of the SAME launch (device-scope loads)
template <bool COHERENT>
__device__ __forceinline__ void ema_embed_row(float *embed_avg, float *embed, const float *embed_sum,
                                              const float *weight, const float *denom, int C, int D,
                                              float omd, int cosine, int do_lerp, int do_update, int c)
{
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    const float w = weight ? omd * weight[c] : omd;
    const float den = do_update ? denom[c] : 1.f;
    float e[8];
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int d = lane + 64 * q;
        e[q] = 0.f;
        if (d < D) {
            const size_t o = (size_t)c * D + d;
            float ea = embed_avg[o];
            if (do_lerp) {
                const float es = COHERENT ? __hip_atomic_load(&embed_sum[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : embed_sum[o];
                ea = aten_lerp(ea, es, w);
                embed_avg[o] = ea;
            }
            if (do_update) {
                e[q] = ea / den;
                ss += e[q] * e[q];
            }
        }
    }
    if (!do_update) return;
    float inv = 1.f;
    if (cosine) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        inv = fmaxf(sqrtf(ss), 1e-6f);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int d = lane + 64 * q;
        if (d < D) embed[(size_t)c * D + d] = cosine ? (e[q] / inv) : e[q];
    }
}

__global__ void __launch_bounds__(256) vq_ema_embed_kernel(float *embed_avg, float *embed, const float *embed_sum,
                                                           const float *weight, const float *denom, int C, int D,
                                                           float omd, int cosine, int do_lerp, int do_update, int64_t hs_sum)
{
    const size_t h = blockIdx.y;           // batched heads: embed_avg / embed [H, C, D], denom [H, C], embed_sum at hs_sum floats
    ema_embed_row<false>(embed_avg + h * C * D, embed + h * C * D, embed_sum ? embed_sum + h * hs_sum : nullptr, weight, denom ? denom + h * C : nullptr,
                  C, D, omd, cosine, do_lerp, do_update, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// tail of the fused train step: workgroups [0, ceil(C / 4)) fold embed_sum into embed_avg and renormalise embed (the kernel above),
// the LAST workgroup reduces the commitment loss' partials (vq_reduce_kernel's arithmetic: fp64, fixed order)
__global__ void __launch_bounds__(256) vq_step_fold_kernel(float *embed_avg, float *embed, const float *embed_sum, const float *denom,
                                                           int C, int D, float omd, int cosine, const double *__restrict__ partials,
                                                           int64_t n_partials, double scale, float *loss_out)
{
    if (blockIdx.x + 1 < gridDim.x) {
        ema_embed_row<false>(embed_avg, embed, embed_sum, nullptr, denom, C, D, omd, cosine, 1, 1, blockIdx.x * 4 + (threadIdx.x >> 6));
        return;
    }
    if (!loss_out) return;
    __shared__ double red[256];
    double sum = 0.0;
    for (int64_t i = threadIdx.x; i < n_partials; i += 256) sum += partials[i];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = (float)(red[0] * scale);
}

extern "C" int vqhip_ema_finalize(float *cluster_size, float *embed_avg, float *embed,
                                  const float *count, const float *embed_sum, const float *weight,
                                  int C, int D, float one_minus_decay, float eps, int cosine,
                                  int do_lerp, int do_update_ema, float *denom_ws, void *stream)
{
    if (!cluster_size || !embed_avg || !embed || C <= 0) VQ_FAIL(VQHIP_EINVAL, "ema_finalize: null pointer or C <= 0");
    if (D < 1 || D > VQ_WIDE_MAX_D) VQ_FAIL(VQHIP_EDIM, "ema_finalize: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (do_lerp && (!count || !embed_sum)) VQ_FAIL(VQHIP_EINVAL, "ema_finalize: do_lerp needs count and embed_sum");
    if (do_update_ema && !denom_ws) VQ_FAIL(VQHIP_EINVAL, "ema_finalize: do_update_ema needs denom_ws");
    hipStream_t st = (hipStream_t)stream;
    if (do_lerp)
        hipLaunchKernelGGL(vq_ema_cs_lerp_kernel, dim3((C + 255) / 256), dim3(256), 0, st, cluster_size, count, weight, C, one_minus_decay, (int64_t)0, (int64_t)0);
    if (do_update_ema) {
        const float ceps = (float)((double)C * (double)eps);
        hipLaunchKernelGGL(vq_ema_denom_kernel, dim3(1), dim3(256), (C <= 16384 ? (size_t)C * 4 : 0), st, cluster_size, C, eps, ceps, denom_ws);
    }
    if (vq_is_wide(D)) {
        if (!(do_lerp || do_update_ema)) return launch_status("vq_ema_finalize");
        return vq_wide_ema_embed(embed_avg, embed, embed_sum, weight, denom_ws, 1, C, D, one_minus_decay, cosine, do_lerp, do_update_ema, 0, stream);
    }
    if (do_lerp || do_update_ema)
        hipLaunchKernelGGL(vq_ema_embed_kernel, dim3((C + 3) / 4), dim3(256), 0, st, embed_avg, embed, embed_sum, weight,
                           denom_ws, C, D, one_minus_decay, cosine, do_lerp, do_update_ema, (int64_t)0);
    return launch_status("vq_ema_finalize");
}

// H codebooks' folds in three launches (blockIdx.y = head): cluster_size [H, C], embed_avg / embed [H, C, D] -- the module buffers of a
// multi-head Codebook as they are -- and stats [H, stats_stride] = embed_sum [C, D] || count [C] per head (vqhip_ema_accumulate_batched).
extern "C" int vqhip_ema_finalize_batched(float *cluster_size, float *embed_avg, float *embed, const float *stats, int64_t stats_stride,
                                          int H, int C, int D, float one_minus_decay, float eps, int cosine, int do_update_ema,
                                          float *denom_ws, void *stream)
{
    if (!cluster_size || !embed_avg || !embed || !stats || C <= 0 || H < 1) VQ_FAIL(VQHIP_EINVAL, "ema_finalize_batched: bad argument");
    if (D < 1 || D > 512) VQ_FAIL(VQHIP_EDIM, "ema_finalize_batched: D=%d unsupported (1..512)", D);
    if (stats_stride < (int64_t)C * D + C) VQ_FAIL(VQHIP_EINVAL, "ema_finalize_batched: stats_stride smaller than C D + C");
    if (do_update_ema && !denom_ws) VQ_FAIL(VQHIP_EINVAL, "ema_finalize_batched: do_update_ema needs denom_ws [H, C]");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vq_ema_cs_lerp_kernel, dim3((C + 255) / 256, H), dim3(256), 0, st, cluster_size, stats + (size_t)C * D, (const float *)nullptr, C,
                       one_minus_decay, (int64_t)C, stats_stride);
    if (do_update_ema) {
        const float ceps = (float)((double)C * (double)eps);
        hipLaunchKernelGGL(vq_ema_denom_kernel, dim3(1, H), dim3(256), (C <= 16384 ? (size_t)C * 4 : 0), st, cluster_size, C, eps, ceps, denom_ws);
    }
    hipLaunchKernelGGL(vq_ema_embed_kernel, dim3((C + 3) / 4, H), dim3(256), 0, st, embed_avg, embed, stats, (const float *)nullptr, denom_ws, C, D,
                       one_minus_decay, cosine, 1, do_update_ema, stats_stride);
    return launch_status("vq_ema_finalize_batched");
}

// The same three launches for H codebooks whose buffers are SEPARATE allocations -- the layers of a (grouped) residual VQ, each with its
// own cluster_size / embed_avg / embed module buffers (state_dict keys layers.{i}._codebook.*): `table` holds H triples of device
// pointers (cluster_size [C], embed_avg [C, D], embed [C, D]), head h's statistics sit at stats + h * stats_stride.
__global__ void __launch_bounds__(256) vq_ema_cs_lerp_tab_kernel(const uintptr_t *table, const float *stats, int64_t stats_stride, int C, int D, float omd)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float *cs = (float *)table[3 * blockIdx.y];
    const float *count = stats + blockIdx.y * stats_stride + (size_t)C * D;
    cs[c] = aten_lerp(cs[c], count[c], omd);
}

__global__ void __launch_bounds__(256) vq_ema_denom_tab_kernel(const uintptr_t *table, int C, float eps, float ceps, float *denom)
{
    const float *cs_g = (const float *)table[3 * blockIdx.y];
    denom += (size_t)blockIdx.y * C;
    __shared__ float part[32];
    __shared__ float total_s;
    extern __shared__ float cs_lds[];
    const int tid = threadIdx.x;
    const float *cs = cs_g;
    if (C <= 16384) {
        for (int c = tid; c < C; c += 256) cs_lds[c] = cs_g[c];
        __syncthreads();
        cs = cs_lds;
    }
    ema_denom_block<256>(cs, C, eps, ceps, denom, part, &total_s);
}

__global__ void __launch_bounds__(256) vq_ema_embed_tab_kernel(const uintptr_t *table, const float *stats, int64_t stats_stride, const float *denom,
                                                               int C, int D, float omd, int cosine, int do_update)
{
    const size_t h = blockIdx.y;
    ema_embed_row<false>((float *)table[3 * h + 1], (float *)table[3 * h + 2], stats + h * stats_stride, nullptr, denom ? denom + h * C : nullptr,
                  C, D, omd, cosine, 1, do_update, blockIdx.x * 4 + (threadIdx.x >> 6));
}

extern "C" int vqhip_ema_finalize_table(const void *table, const float *stats, int64_t stats_stride, int H, int C, int D,
                                        float one_minus_decay, float eps, int cosine, int do_update_ema, float *denom_ws, void *stream)
{
    if (!table || !stats || C <= 0 || H < 1) VQ_FAIL(VQHIP_EINVAL, "ema_finalize_table: bad argument");
    if (D < 1 || D > 512) VQ_FAIL(VQHIP_EDIM, "ema_finalize_table: D=%d unsupported (1..512)", D);
    if (stats_stride < (int64_t)C * D + C) VQ_FAIL(VQHIP_EINVAL, "ema_finalize_table: stats_stride smaller than C D + C");
    if (do_update_ema && !denom_ws) VQ_FAIL(VQHIP_EINVAL, "ema_finalize_table: do_update_ema needs denom_ws [H, C]");
    hipStream_t st = (hipStream_t)stream;
    const uintptr_t *tab = (const uintptr_t *)table;
    hipLaunchKernelGGL(vq_ema_cs_lerp_tab_kernel, dim3((C + 255) / 256, H), dim3(256), 0, st, tab, stats, stats_stride, C, D, one_minus_decay);
    if (do_update_ema) {
        const float ceps = (float)((double)C * (double)eps);
        hipLaunchKernelGGL(vq_ema_denom_tab_kernel, dim3(1, H), dim3(256), (C <= 16384 ? (size_t)C * 4 : 0), st, tab, C, eps, ceps, denom_ws);
    }
    hipLaunchKernelGGL(vq_ema_embed_tab_kernel, dim3((C + 3) / 4, H), dim3(256), 0, st, tab, stats, stats_stride, denom_ws, C, D, one_minus_decay,
                       cosine, do_update_ema);
    return launch_status("vq_ema_finalize_table");
}

// Renormalisation of ONE SHARD of a codebook partitioned over ranks (parallel.ShardedVectorQuantize): the Laplace smoothing of
// vqp.py:152-154 / :577 uses sum(cluster_size) and the code count of the WHOLE codebook, which the caller all-reduces and passes in
// (a device scalar: no host sync); everything else is update_ema's arithmetic on the shard's rows (vqp.py:576-584).
__global__ void __launch_bounds__(256) vq_ema_denom_ext_kernel(const float *cs, int C, float eps, float ceps_total, const float *total_p, float *denom)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float total = *total_p;
    denom[c] = (cs[c] + eps) / (total + ceps_total) * total;
}

extern "C" int vqhip_ema_renormalize_shard(const float *cluster_size, float *embed_avg, float *embed, int C, int D, float eps,
                                           const float *total_cluster_size, int C_total, int cosine, float *denom_ws, void *stream)
{
    if (!cluster_size || !embed_avg || !embed || !total_cluster_size || !denom_ws || C <= 0 || C_total < C)
        VQ_FAIL(VQHIP_EINVAL, "ema_renormalize_shard: null pointer, C <= 0 or C_total < C");
    if (D < 1 || D > 512) VQ_FAIL(VQHIP_EDIM, "ema_renormalize_shard: D=%d unsupported (1..512)", D);
    hipStream_t st = (hipStream_t)stream;
    const float ceps = (float)((double)C_total * (double)eps);
    hipLaunchKernelGGL(vq_ema_denom_ext_kernel, dim3((C + 255) / 256), dim3(256), 0, st, cluster_size, C, eps, ceps, total_cluster_size, denom_ws);
    hipLaunchKernelGGL(vq_ema_embed_kernel, dim3((C + 3) / 4), dim3(256), 0, st, embed_avg, embed, (const float *)nullptr, (const float *)nullptr,
                       denom_ws, C, D, 0.f, cosine, 0, 1, (int64_t)0);
    return launch_status("vq_ema_renormalize_shard");
}

// A codebook shared by the Q stages of a residual VQ is lerp-ed Q times in stage order (rvq.py:213-217 + vqp.py:616-617: every
// stage's update_codebook folds its statistics; the renormalisation runs once at the end, rvq.py:593-598).  The lerps are
// element-wise, so one launch applies all Q of them -- the same aten_lerp calls in the same order, bit for bit -- instead of
// 2 Q launches.  stats: Q blocks of `stride` floats, each embed_sum [C, D] followed by count [C].
__global__ void __launch_bounds__(256) vq_ema_cs_lerp_many_kernel(float *cs, const float *stats, int Q, int64_t stride, int C, int D, float omd)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float v = cs[c];
    for (int q = 0; q < Q; ++q) v = aten_lerp(v, stats[(size_t)q * stride + (size_t)C * D + c], omd);
    cs[c] = v;
}

__global__ void __launch_bounds__(256) vq_ema_embed_many_kernel(float *embed_avg, float *embed, const float *stats, int Q, int64_t stride,
                                                                const float *denom, int C, int D, float omd, int cosine, int do_update)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const float den = do_update ? denom[c] : 1.f;
    float e[8];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int d = lane + 64 * k;
        e[k] = 0.f;
        if (d < D) {
            const size_t o = (size_t)c * D + d;
            float ea = embed_avg[o];
            for (int q = 0; q < Q; ++q) ea = aten_lerp(ea, stats[(size_t)q * stride + o], omd);
            embed_avg[o] = ea;
            if (do_update) {
                e[k] = ea / den;
                ss += e[k] * e[k];
            }
        }
    }
    if (!do_update) return;
    float inv = 1.f;
    if (cosine) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        inv = fmaxf(sqrtf(ss), 1e-6f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int d = lane + 64 * k;
        if (d < D) embed[(size_t)c * D + d] = cosine ? (e[k] / inv) : e[k];
    }
}

extern "C" int vqhip_ema_fold_many(float *cluster_size, float *embed_avg, float *embed, const float *stats, int Q, int64_t stride,
                                   int C, int D, float one_minus_decay, float eps, int cosine, int do_update_ema, float *denom_ws,
                                   void *stream)
{
    if (!cluster_size || !embed_avg || !embed || !stats || C <= 0 || Q < 1) VQ_FAIL(VQHIP_EINVAL, "ema_fold_many: bad argument");
    if (D < 1 || D > 512) VQ_FAIL(VQHIP_EDIM, "ema_fold_many: D=%d unsupported (1..512)", D);
    if (stride < (int64_t)C * D + C) VQ_FAIL(VQHIP_EINVAL, "ema_fold_many: stride smaller than C D + C");
    if (do_update_ema && !denom_ws) VQ_FAIL(VQHIP_EINVAL, "ema_fold_many: do_update_ema needs denom_ws");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vq_ema_cs_lerp_many_kernel, dim3((C + 255) / 256), dim3(256), 0, st, cluster_size, stats, Q, stride, C, D, one_minus_decay);
    if (do_update_ema) {
        const float ceps = (float)((double)C * (double)eps);
        hipLaunchKernelGGL(vq_ema_denom_kernel, dim3(1), dim3(256), (C <= 16384 ? (size_t)C * 4 : 0), st, cluster_size, C, eps, ceps, denom_ws);
    }
    hipLaunchKernelGGL(vq_ema_embed_many_kernel, dim3((C + 3) / 4), dim3(256), 0, st, embed_avg, embed, stats, Q, stride, denom_ws, C, D,
                       one_minus_decay, cosine, do_update_ema);
    return launch_status("vq_ema_fold_many");
}

// ------------------------------------------------------------------------------------------------
// dead-code replacement on the device (vqp.py:544-574) and the k-means centroid update (vqp.py:262-276)
// ------------------------------------------------------------------------------------------------
// expire: the j-th expired code (cluster_size < threshold, ascending code order -- the order in which the reference's boolean-mask
// assignment `embed[mask] = sampled` consumes `sampled`) takes candidate row j: embed[c] = cand[j], cluster_size[c] = reset,
// embed_avg[c] = cand[j] * reset.  The candidates are rows drawn by the caller from torch's generator (randperm(n)[:C] like
// vqp.py:156-163, l2-normalised for the cosine metric); nothing returns to the host, so the whole step can be captured in a HIP
// graph.  One workgroup of 1024 threads: the rank of a code among the expired ones is a block-wide prefix count.
__global__ void __launch_bounds__(1024) vq_expire_kernel(float *__restrict__ cluster_size, float *__restrict__ embed_avg,
                                                         float *__restrict__ embed, const float *__restrict__ cand, int C, int D,
                                                         float threshold, float reset, int *__restrict__ n_expired_out)
{
    __shared__ int wsum[16];
    __shared__ int base_sh, n_chunk;
    __shared__ int ex_code[1024], ex_rank[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int c0 = 0; c0 < C; c0 += 1024) {
        const int c = c0 + tid;
        const bool ex = c < C && cluster_size[c] < threshold;
        const unsigned long long bal = __ballot(ex);
        const int within = (int)__popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = (int)__popcll(bal);
        __syncthreads();
        int before = base_sh;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        const int rank = before + within;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base_sh += t; }
        if (ex) cluster_size[c] = reset;
        // the row copies: every expired code of this chunk, D floats each, spread over the block below
        if (tid == 0) n_chunk = 0;
        __syncthreads();
        if (ex) { const int s = atomicAdd(&n_chunk, 1); ex_code[s] = c; ex_rank[s] = rank; }
        __syncthreads();
        const int nc = n_chunk;
        for (int p = tid; p < nc * D; p += 1024) {
            const int s = p / D, k = p - s * D;
            const float v = cand[(size_t)ex_rank[s] * D + k];
            embed[(size_t)ex_code[s] * D + k] = v;
            embed_avg[(size_t)ex_code[s] * D + k] = v * reset;
        }
        __syncthreads();
    }
    if (tid == 0 && n_expired_out) *n_expired_out = base_sh;
}

extern "C" int vqhip_expire_scatter(float *cluster_size, float *embed_avg, float *embed, const float *candidates, int C, int D,
                                    float threshold, float reset, int *n_expired_out, void *stream)
{
    if (!cluster_size || !embed_avg || !embed || !candidates || C <= 0 || D <= 0) VQ_FAIL(VQHIP_EINVAL, "expire_scatter: bad argument");
    hipLaunchKernelGGL(vq_expire_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cluster_size, embed_avg, embed, candidates, C, D,
                       threshold, reset, n_expired_out);
    return launch_status("vq_expire_kernel");
}

// Dead-code replacement without a candidate tensor (vqp.py:544-574 with sample_vectors :180-188: expired codes take batch rows drawn
// without replacement): code c takes row pi(c) of the batch, pi = an affine permutation t -> (a t + b) mod p of Z_p, p the
// smallest prime >= n, cycle-walked into [0, n) (a bijection of [0, n), so distinct codes take distinct rows whenever C <= n; fewer
// rows than codes wrap around, the reference's with-replacement case).  (a, b): two draws of the caller's generator, read from
// device memory -- the whole replacement is this one launch, nothing depends on how many codes expired.  One wave per code.
__global__ void __launch_bounds__(256) vq_expire_pick_kernel(float *__restrict__ cluster_size, float *__restrict__ embed_avg,
                                                             float *__restrict__ embed, const void *__restrict__ rows, int bf16,
                                                             int64_t n, int64_t ldx, const int64_t *__restrict__ ab, int64_t p,
                                                             int C, int D, float threshold, float reset, int cosine)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    if (!(cluster_size[c] < threshold)) return;
    const unsigned long long a = (unsigned long long)ab[0], b = (unsigned long long)ab[1], pp = (unsigned long long)p;
    unsigned long long pick = (a * ((unsigned long long)c % pp) + b) % pp;
    for (int it = 0; it < 64 && pick >= (unsigned long long)n; ++it) pick = (a * pick + b) % pp;
    if (pick >= (unsigned long long)n) pick %= (unsigned long long)n;       // (a short cycle of pi inside [n, p): practically never)
    // The affine map alone sends neighbouring codes to rows a apart (an arithmetic progression: on a flattened [b, n, d] batch that
    // is e.g. the same position of consecutive sequences -- ADVICE r4).  A 4-round Feistel network on the smallest even-width
    // power-of-two domain >= n, keyed by (a, b) and cycle-walked into [0, n), permutes the rows once more: a composition of two
    // permutations of [0, n), so distinct codes still take distinct rows; the picks no longer form a progression.
    if (n > 2) {
        int kb = 1;
        while (((unsigned long long)1 << kb) < (unsigned long long)n) ++kb;
        kb += kb & 1;
        const int hb = kb >> 1;
        const unsigned mask = (1u << hb) - 1u;
        unsigned long long v = pick;
        for (int it = 0; it < 64; ++it) {
            unsigned L = (unsigned)(v >> hb) & mask, R = (unsigned)v & mask;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned f = R * 0x9E3779B1u + (unsigned)(r & 1 ? b : a) + 0x85EBCA6Bu * (unsigned)(r + 1) + (unsigned)((r & 2 ? a : b) >> 17);
                f ^= f >> 15; f *= 0x2C1B3C6Du; f ^= f >> 12; f *= 0x297A2D39u; f ^= f >> 15;
                const unsigned t = L ^ (f & mask);
                L = R; R = t;
            }
            v = ((unsigned long long)L << hb) | R;
            if (v < (unsigned long long)n) break;
        }
        if (v < (unsigned long long)n) pick = v;          // (64 walks without landing in [0, n) -- domain < 4 n, probability 4^-64: keep the affine pick)
    }
    auto ld = [&](int k) { return bf16 ? bf16_bits_to_f32(((const unsigned short *)rows)[pick * ldx + k]) : ((const float *)rows)[pick * ldx + k]; };
    float inv = 1.f;
    if (cosine) {                                                           // l2norm of the sampled rows (vqp.py:545-546, 37-38)
        float ss = 0.f;
        for (int k = lane; k < D; k += 64) { const float v = ld(k); ss += v * v; }     // (lane l: elements l, l + 64, ... as before)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        inv = fmaxf(sqrtf(ss), 1e-6f);
    }
    for (int k = lane; k < D; k += 64) {                                    // any D (the row is re-read: at most 8 KiB)
        const float v = ld(k);
        const float e = cosine ? v / inv : v;
        embed[(size_t)c * D + k] = e;
        embed_avg[(size_t)c * D + k] = e * reset;
    }
    if (lane == 0) cluster_size[c] = reset;
}

extern "C" int vqhip_expire_pick(float *cluster_size, float *embed_avg, float *embed, const void *rows, int x_dtype, int64_t n, int64_t ldx,
                                 const int64_t *ab, int64_t p, int C, int D, float threshold, float reset, int cosine, void *stream)
{
    if (!cluster_size || !embed_avg || !embed || !rows || !ab || C <= 0) VQ_FAIL(VQHIP_EINVAL, "expire_pick: bad argument");
    if (D < 1 || D > VQ_WIDE_MAX_D) VQ_FAIL(VQHIP_EDIM, "expire_pick: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (x_dtype != VQHIP_F32 && x_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "expire_pick: unknown dtype");
    if (n < 1 || p < n || p < 2 || p > 0x7fffffffLL || ldx < D) VQ_FAIL(VQHIP_EINVAL, "expire_pick: need 1 <= n <= p < 2^31 (p prime), ldx >= D");
    hipLaunchKernelGGL(vq_expire_pick_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, cluster_size, embed_avg, embed, rows,
                       x_dtype == VQHIP_BF16 ? 1 : 0, n, ldx, ab, p, C, D, threshold, reset, cosine);
    return launch_status("vq_expire_pick_kernel");
}

// k-means centroid update of one iteration (vqp.py:262-276): means[c] = embed_sum[c] / count[c] where count[c] > 0 (l2-normalised for
// the cosine metric, eps 1e-6 as vqp.py:37-38), unchanged where the bin is empty.  In place on `means`.
__global__ void __launch_bounds__(256) vq_kmeans_update_kernel(float *__restrict__ means, const float *__restrict__ embed_sum,
                                                               const float *__restrict__ count, int C, int D, int cosine)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const float n = count[c];
    if (n == 0.f) return;
    float ss = 0.f;
    for (int k = lane; k < D; k += 64) { const float v = embed_sum[(size_t)c * D + k] / n; ss += v * v; }
    float inv = 1.f;
    if (cosine) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        inv = 1.f / fmaxf(sqrtf(ss), 1e-6f);
    }
    for (int k = lane; k < D; k += 64) {
        const float v = embed_sum[(size_t)c * D + k] / n;
        means[(size_t)c * D + k] = cosine ? v * inv : v;
    }
}

extern "C" int vqhip_kmeans_update(float *means, const float *embed_sum, const float *count, int C, int D, int cosine, void *stream)
{
    if (!means || !embed_sum || !count || C <= 0 || D <= 0) VQ_FAIL(VQHIP_EINVAL, "kmeans_update: bad argument");
    hipLaunchKernelGGL(vq_kmeans_update_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, means, embed_sum, count, C, D, cosine);
    return launch_status("vq_kmeans_update_kernel");
}

// ------------------------------------------------------------------------------------------------
// K11 helper of the codebook-sharded argmin (vector_quantize_pytorch_amd/parallel.py): the per-shard winner
// (score, global index) as ONE order-preserving int64 key, so that a single all_reduce(MAX) of N x 8 bytes picks the global
// winner with ATen argmax's first-occurrence rule (vqp.py:140).  High word: the IEEE-754 bits of the score TO MAXIMISE mapped
// to a signed int32 that sorts like the float (negative floats: magnitude bits flipped); low word: 0xFFFFFFFF - index, so that
// among equal scores the LOWEST index has the LARGEST key.  (torch has no uint64 collectives, hence the signed construction.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vq_pack_best_kernel(const float *__restrict__ best, const int64_t *__restrict__ idx, int64_t N,
                                                           int64_t offset, int negate, int64_t *__restrict__ key)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = best[n];
    if (negate) s = -s;
    int b = __float_as_int(s);
    b = b < 0 ? (b ^ 0x7FFFFFFF) : b;
    const int64_t g = idx[n] + offset;
    key[n] = (int64_t)(((uint64_t)(uint32_t)b << 32) | (uint64_t)(uint32_t)(0xFFFFFFFFu - (uint32_t)g));
}

__global__ void __launch_bounds__(256) vq_unpack_best_kernel(const int64_t *__restrict__ key, int64_t N, int64_t lo, int64_t hi, int negate,
                                                             int64_t *__restrict__ gidx, int64_t *__restrict__ local, float *__restrict__ best)
{
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const uint64_t k = (uint64_t)key[n];
    const int64_t g = (int64_t)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull));
    int b = (int)(uint32_t)(k >> 32);
    b = b < 0 ? (b ^ 0x7FFFFFFF) : b;
    if (gidx) gidx[n] = g;
    if (local) local[n] = (g >= lo && g < hi) ? g - lo : -1;       // -1: another rank owns the winner (skipped by decode / EMA)
    if (best) { const float s = __int_as_float(b); best[n] = negate ? -s : s; }
}

extern "C" int vqhip_pack_best(const float *best, const int64_t *idx, int64_t N, int64_t index_offset, int negate, int64_t *key_out,
                               void *stream)
{
    if (N < 0 || (N > 0 && (!best || !idx || !key_out)) || index_offset < 0) VQ_FAIL(VQHIP_EINVAL, "pack_best: bad argument");
    if (N == 0) return 0;
    hipLaunchKernelGGL(vq_pack_best_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, best, idx, N,
                       index_offset, negate, key_out);
    return launch_status("vq_pack_best_kernel");
}

extern "C" int vqhip_unpack_best(const int64_t *key, int64_t N, int64_t own_lo, int64_t own_hi, int negate, int64_t *gidx_out,
                                 int64_t *local_out, float *best_out, void *stream)
{
    if (N < 0 || (N > 0 && !key) || own_lo > own_hi) VQ_FAIL(VQHIP_EINVAL, "unpack_best: bad argument");
    if (N == 0) return 0;
    hipLaunchKernelGGL(vq_unpack_best_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, key, N, own_lo,
                       own_hi, negate, gidx_out, local_out, best_out);
    return launch_status("vq_unpack_best_kernel");
}

// ------------------------------------------------------------------------------------------------
// decode: out[n,:] = sum_q embed_q[idx[n,q],:]   (sequential in q, like the reference's running sum)
// ------------------------------------------------------------------------------------------------
// one wave per output row.  The Q indices of a row are fetched first, then the code rows of 8 stages at a time are all
// requested before any is added (the first version chained index load -> row load -> add per stage: latency bound at
// 1.1 TB/s of output); the adds keep the stage order, so the result is the same running sum as rvq.py:525.
// idx_stride: elements between the rows of idx (>= Q: the first Q columns of a wider index tensor); accumulate != 0 (fp32 outputs): the
// running sum starts from what `out` holds -- a decode split into stage ranges continues the SAME left-to-right sum of rvq.py:525
template <bool VEC>
__global__ void __launch_bounds__(256) vq_decode_kernel(const int64_t *__restrict__ idx, int64_t idx_stride, int64_t N, int Q,
                                                        const float *__restrict__ embed, int64_t qstride, int C, int D,
                                                        void *out, int out_bf16, int64_t ldo, int accumulate)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    constexpr int U = 8;
    if (VEC) {                      // D % 4 == 0, 16-byte aligned rows: lane owns columns 4 lane + 256 h .. +3
        f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (accumulate) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = lane * 4 + 256 * h;
                if (d < D) s[h] = *(const f32x4 *)((const float *)out + n * ldo + d);
            }
        }
        for (int q0 = 0; q0 < Q; q0 += U) {
            int64_t c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) c[u] = (q0 + u < Q) ? idx[n * idx_stride + q0 + u] : -1;
            f32x4 v[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = c[u] >= 0 && c[u] < C;
                const float *r = embed + (size_t)(q0 + u < Q ? q0 + u : 0) * qstride + (size_t)(ok ? c[u] : 0) * D;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = lane * 4 + 256 * h;
                    v[u][h] = (ok && d < D) ? *(const f32x4 *)(r + d) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = c[u] >= 0 && c[u] < C;      // a skipped stage adds nothing (not even +0 to a -0)
                if (ok) { s[0] += v[u][0]; s[1] += v[u][1]; }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = lane * 4 + 256 * h;
            if (d < D) {
                if (out_bf16) {
                    uint2 w;
                    w.x = (unsigned)f32_to_bf16_rne(s[h].x) | ((unsigned)f32_to_bf16_rne(s[h].y) << 16);
                    w.y = (unsigned)f32_to_bf16_rne(s[h].z) | ((unsigned)f32_to_bf16_rne(s[h].w) << 16);
                    *(uint2 *)((unsigned short *)out + n * ldo + d) = w;
                } else {
                    *(f32x4 *)((float *)out + n * ldo + d) = s[h];
                }
            }
        }
        return;
    }
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int d = lane + 64 * u; s[u] = (accumulate && d < D) ? ((const float *)out)[n * ldo + d] : 0.f; }
    for (int q = 0; q < Q; ++q) {
        const int64_t c = idx[n * idx_stride + q];
        if (c < 0 || c >= C) continue;
        const float *r = embed + (size_t)q * qstride + (size_t)c * D;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int d = lane + 64 * u;
            if (d < D) s[u] += r[d];
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int d = lane + 64 * u;
        if (d < D) {
            if (out_bf16) ((unsigned short *)out)[n * ldo + d] = f32_to_bf16_rne(s[u]);
            else ((float *)out)[n * ldo + d] = s[u];
        }
    }
}

// One codebook shared by all stages and small enough that a 32-column slice of it fits in LDS (C <= 1024: 128 KiB): the
// row-per-wave kernel above gathers Q code rows per output row from L2 (cfg 3: 8 KiB gathered per 1 KiB written, 1.4 TB/s of
// output); here a workgroup parks columns [32 s, 32 s + 32) of every code in LDS once and then streams its share of the rows --
// 8 lanes per row, 16 bytes each, the Q gathers come from LDS (conflict-free: consecutive lanes, consecutive 16-byte pieces of a
// 128-byte row), the sums run in stage order like the reference's (rvq.py:525), and every store is a whole 128-byte line.
#define VQ_DECODE_LDS_COLS 32
#ifndef VQ_DECODE_LDS_PAD
#define VQ_DECODE_LDS_PAD 0           // floats of padding behind a code's 32-column piece in LDS.  rocprofv3 counts bank conflicts on 90 % of
                                      // this kernel's LDS cycles (eight random 128-byte pieces per wave read: even codes on banks 0 .. 31, odd
                                      // ones on 32 .. 63); 4 / 8 floats of padding measured 105 / 100 us against 104 (cfg-3 shape): not the bound
#endif
#define VQ_DECODE_LDS_ROW (VQ_DECODE_LDS_COLS + VQ_DECODE_LDS_PAD)
__global__ void __launch_bounds__(1024) vq_decode_lds_kernel(const int64_t *__restrict__ idx, int64_t N, int Q, const float *__restrict__ embed,
                                                             int C, int D, void *out, int out_bf16, int64_t ldo, int rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int nsl = D / VQ_DECODE_LDS_COLS;
    const int sl = blockIdx.x % nsl;                       // column slice
    const int64_t rb = blockIdx.x / nsl;                   // row chunk
    // codes: 8 lanes x 16 bytes per code row slice
    for (int i = tid; i < C * 8; i += blockDim.x)
        *(f32x4 *)(smem + ((size_t)(i >> 3) * VQ_DECODE_LDS_ROW + (i & 7) * 4) * 4) = *(const f32x4 *)(embed + (size_t)(i >> 3) * D + sl * VQ_DECODE_LDS_COLS + (i & 7) * 4);
    __syncthreads();
    const int part = tid & 7;                              // which 16 bytes of the slice
    const int64_t r0 = rb * rows_per_block, r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
    for (int64_t n = r0 + (tid >> 3); n < r1; n += blockDim.x >> 3) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < Q; q0 += 8) {
            // lane `part` of the row's group fetches stage q0 + part's index; the group shares them by shuffle
            const int64_t mine = (q0 + part < Q) ? idx[n * Q + q0 + part] : -1;
            const int lo = (int)mine;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = __shfl(lo, (threadIdx.x & 63 & ~7) + u, 64);
                if (q0 + u < Q && c >= 0 && c < C) s += *(const f32x4 *)(smem + ((size_t)c * VQ_DECODE_LDS_ROW + part * 4) * 4);   // a skipped stage adds nothing
            }
        }
        const int d = sl * VQ_DECODE_LDS_COLS + part * 4;
        if (out_bf16) {
            uint2 w;
            w.x = (unsigned)f32_to_bf16_rne(s.x) | ((unsigned)f32_to_bf16_rne(s.y) << 16);
            w.y = (unsigned)f32_to_bf16_rne(s.z) | ((unsigned)f32_to_bf16_rne(s.w) << 16);
            *(uint2 *)((unsigned short *)out + n * ldo + d) = w;
        } else {
            *(f32x4 *)((float *)out + n * ldo + d) = s;
        }
    }
}

extern "C" int vqhip_decode_sum(const int64_t *idx, int64_t N, int Q, const float *embed, int64_t embed_qstride,
                                int C, int D, void *out, int out_dtype, int64_t ldo, void *stream)
{
    return vqhip_decode_sum_range(idx, Q, N, Q, embed, embed_qstride, C, D, out, out_dtype, ldo, 0, stream);
}

extern "C" int vqhip_decode_sum_range(const int64_t *idx, int64_t idx_stride, int64_t N, int Q, const float *embed, int64_t embed_qstride,
                                      int C, int D, void *out, int out_dtype, int64_t ldo, int accumulate, void *stream)
{
    if (N < 0 || Q < 1 || C <= 0 || idx_stride < Q) VQ_FAIL(VQHIP_EINVAL, "decode_sum: bad size");
    if (N == 0) return 0;
    if (!idx || !embed || !out) VQ_FAIL(VQHIP_EINVAL, "decode_sum: null pointer");
    if (accumulate && out_dtype != VQHIP_F32) VQ_FAIL(VQHIP_EINVAL, "decode_sum_range: accumulate continues an fp32 running sum");
    if ((accumulate || idx_stride != Q) && vq_is_wide(D)) VQ_FAIL(VQHIP_EDIM, "decode_sum_range: stage ranges are for D <= 512");
    if (D < 1 || D > VQ_WIDE_MAX_D) VQ_FAIL(VQHIP_EDIM, "decode_sum: D=%d unsupported (1..%d)", D, VQ_WIDE_MAX_D);
    if (out_dtype != VQHIP_F32 && out_dtype != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "decode_sum: unknown dtype");
    if (vq_is_wide(D)) return vq_wide_decode_sum(idx, N, Q, embed, embed_qstride, C, D, out, out_dtype, ldo, stream);
    const int oes = (out_dtype == VQHIP_BF16) ? 2 : 4;
    const bool vec = (D % 4 == 0) && (embed_qstride % 4 == 0) && ((((uintptr_t)embed) & 15) == 0) &&
                     ((((uintptr_t)out) % (4 * oes)) == 0) && ((ldo * oes) % (4 * oes) == 0);
    if (vec && Q >= 2 && (embed_qstride == 0 || Q == 1) && C <= 1024 && D % VQ_DECODE_LDS_COLS == 0 && N >= 16384 && !accumulate && idx_stride == Q) {
        // shared codebook, several stages: the codes' column slices live in LDS (the gathers from L2 were the bound)
        const int smem = C * VQ_DECODE_LDS_ROW * 4;
        static VqAttrOnce once;
        if (int rc = vq_set_max_smem(once, (const void *)vq_decode_lds_kernel, 1024 * VQ_DECODE_LDS_ROW * 4, "vq_decode_lds_kernel")) return rc;
        const int nsl = D / VQ_DECODE_LDS_COLS;
        // about four workgroups per CU and slice-group; every workgroup re-reads its slice of the codebook (C * 128 bytes)
        int64_t chunks = (256 * 4) / nsl; if (chunks < 1) chunks = 1;
        int64_t rpb = (N + chunks - 1) / chunks; rpb = (rpb + 127) / 128 * 128;
        chunks = (N + rpb - 1) / rpb;
        hipLaunchKernelGGL(vq_decode_lds_kernel, dim3((unsigned)(chunks * nsl)), dim3(1024), smem, (hipStream_t)stream, idx, N, Q, embed, C, D,
                           out, out_dtype == VQHIP_BF16, ldo, (int)rpb);
        return launch_status("vq_decode_lds_kernel");
    }
    if (vec)
        hipLaunchKernelGGL(vq_decode_kernel<true>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, idx, idx_stride, N, Q, embed,
                           embed_qstride, C, D, out, out_dtype == VQHIP_BF16, ldo, accumulate);
    else
        hipLaunchKernelGGL(vq_decode_kernel<false>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, idx, idx_stride, N, Q, embed,
                           embed_qstride, C, D, out, out_dtype == VQHIP_BF16, ldo, accumulate);
    return launch_status("vq_decode_kernel");
}

// ------------------------------------------------------------------------------------------------
// channel-first callers (channel_last = False, accept_image_fmap: vqp.py:1136-1147 rearranges 'b d n -> b n d' and back): batched
// transposing copy in [B, R, S] (batches in_bstride elements apart: a channel group of a wider map) -> out [B, S, R], one LDS tile of 256 bytes x 256 bytes per workgroup, 16-byte accesses on both
// sides (ATen's strided copy reads 4-byte elements 4 R bytes apart: 1.6 TB/s on a 1 GB tensor; this: HBM speed).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) vq_transpose_kernel(const T *__restrict__ in, T *__restrict__ out, int R, int S, int64_t in_bstride, int vec_ok)
{
    constexpr int VEC = 16 / (int)sizeof(T);      // elements per 16-byte access
    constexpr int TILE = 256 / (int)sizeof(T);    // 64 (4-byte elements) / 128 (2-byte elements)
    constexpr int TPR = TILE / VEC;               // threads per tile row: 16
    constexpr int RPP = 256 / TPR;                // tile rows per pass: 16
    constexpr int LD = TILE + VEC + (sizeof(T) == 2 ? 2 : 1);   // padded row (odd number of 4-byte words: column reads spread over the banks)
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    T *tile = (T *)smem_t;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.z;
    const int r0 = blockIdx.y * TILE, s0 = blockIdx.x * TILE;
    const T *src = in + b * in_bstride;
    T *dst = out + b * (int64_t)R * S;
    const int tr = tid / TPR, tc = (tid % TPR) * VEC;
    const bool full = vec_ok && (r0 + TILE <= R) && (s0 + TILE <= S);
    if (full) {
#pragma unroll
        for (int p = 0; p < TILE / RPP; ++p) {
            const int r = p * RPP + tr;
            const uint4 v = *(const uint4 *)(src + (int64_t)(r0 + r) * S + s0 + tc);
            const T *e = (const T *)&v;
#pragma unroll
            for (int i = 0; i < VEC; ++i) tile[r * LD + tc + i] = e[i];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < TILE / RPP; ++p) {
            const int sl = p * RPP + tr;          // output row (an s) of this tile
            uint4 v;
            T *e = (T *)&v;
#pragma unroll
            for (int i = 0; i < VEC; ++i) e[i] = tile[(tc + i) * LD + sl];
            *(uint4 *)(dst + (int64_t)(s0 + sl) * R + r0 + tc) = v;
        }
        return;
    }
    for (int p = 0; p < TILE / RPP; ++p) {
        const int r = p * RPP + tr;
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            if (r0 + r < R && s0 + tc + i < S) tile[r * LD + tc + i] = src[(int64_t)(r0 + r) * S + s0 + tc + i];
    }
    __syncthreads();
    for (int p = 0; p < TILE / RPP; ++p) {
        const int sl = p * RPP + tr;
#pragma unroll
        for (int i = 0; i < VEC; ++i)
            if (s0 + sl < S && r0 + tc + i < R) dst[(int64_t)(s0 + sl) * R + r0 + tc + i] = tile[(tc + i) * LD + sl];
    }
}

extern "C" int vqhip_transpose_batched(const void *in, void *out, int elem_bytes, int64_t B, int64_t R, int64_t S, int64_t in_bstride, void *stream)
{
    if (B < 0 || R < 0 || S < 0) VQ_FAIL(VQHIP_EINVAL, "transpose_batched: negative size");
    if (B == 0 || R == 0 || S == 0) return 0;
    if (!in || !out) VQ_FAIL(VQHIP_EINVAL, "transpose_batched: null pointer");
    if (elem_bytes != 2 && elem_bytes != 4) VQ_FAIL(VQHIP_EINVAL, "transpose_batched: elem_bytes %d (2 or 4)", elem_bytes);
    if (R > 0x7fffffff || S > 0x7fffffff || B > 65535) VQ_FAIL(VQHIP_EDIM, "transpose_batched: R, S < 2^31, B <= 65535");
    const int tile = 256 / elem_bytes, vec = 16 / elem_bytes;
    if (in_bstride < R * S) VQ_FAIL(VQHIP_EINVAL, "transpose_batched: batch stride smaller than R * S");
    const int vec_ok = ((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 15) == 0 && (R % vec) == 0 && (S % vec) == 0 && (in_bstride % vec) == 0) ? 1 : 0;
    const dim3 grid((unsigned)((S + tile - 1) / tile), (unsigned)((R + tile - 1) / tile), (unsigned)B);
    if (grid.y > 65535) VQ_FAIL(VQHIP_EDIM, "transpose_batched: R too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (elem_bytes == 4) {
        const size_t lds = (size_t)64 * (64 + 4 + 1) * 4;
        hipLaunchKernelGGL(vq_transpose_kernel<unsigned>, grid, dim3(256), lds, st, (const unsigned *)in, (unsigned *)out, (int)R, (int)S, in_bstride, vec_ok);
    } else {
        const size_t lds = (size_t)128 * (128 + 8 + 2) * 2;
        hipLaunchKernelGGL(vq_transpose_kernel<unsigned short>, grid, dim3(256), lds, st, (const unsigned short *)in, (unsigned short *)out,
                           (int)R, (int)S, in_bstride, vec_ok);
    }
    return launch_status("vq_transpose_kernel");
}

// ------------------------------------------------------------------------------------------------
// fused train step (vqhip_vq_train_step): the launches of pack -> search -> statistics -> fold with everything that only exists
// because they are separate API calls removed -- ONE zeroing kernel instead of four memsets / fills, cluster_size folded by the scan
// kernel, embed_avg / embed / loss by one tail kernel.  (Counting the rows per code inside the search -- one global atomic per
// certified row -- was built and measured: it saves the 14 us histogram pass and costs the search 20 us; not kept.)
// ------------------------------------------------------------------------------------------------

static inline size_t step_ws_screen(int64_t N) { return align_up(vqhip_screen_workspace_bytes(N), 256); }

// rows per chunk of a pipelined step: whole screening workgroups (256 rows)
extern "C" int64_t vqhip_vq_step_chunk_rows(int64_t N, int chunks)
{
    if (N <= 0) return 0;
    const int64_t K = chunks < 1 ? 1 : (chunks > VQ_STEP_MAX_CHUNKS ? VQ_STEP_MAX_CHUNKS : chunks);
    return ((N + K - 1) / K + 255) / 256 * 256;
}

// workspace of a K-chunk step: per chunk [screening workspace | statistics workspace (histogram first)], then the denominators
// [C], then the squared-error partials of all chunks (contiguous: one reduction).  Sized for the worst K <= VQ_STEP_MAX_CHUNKS
// (the row-sized parts do not depend on K beyond alignment; the C-sized parts are counted once per chunk).
static inline size_t step_ws_total(int64_t N, int C, int K)
{
    const int64_t rpc = vqhip_vq_step_chunk_rows(N, K);
    size_t tot = 0, parts = 0;
    for (int64_t r0 = 0; r0 < N; r0 += rpc) {
        const int64_t n = N - r0 < rpc ? N - r0 : rpc;
        tot += step_ws_screen(n) + align_up(vqhip_ema_workspace_bytes(n, C), 256);
        parts += (size_t)seg_work_items(n, C);
    }
    return tot + align_up((size_t)C * 4, 256) + align_up(parts * sizeof(double), 256);
}

extern "C" size_t vqhip_vq_step_workspace_bytes(int64_t N, int C)
{
    if (N <= 0 || C <= 0) return 0;
    size_t worst = 0;
    for (int K = 1; K <= VQ_STEP_MAX_CHUNKS; ++K) {
        const size_t t = step_ws_total(N, C, K);
        worst = t > worst ? t : worst;
    }
    return worst;
}

extern "C" int vqhip_vq_step_supported(int x_dtype, int64_t N, int D, int C)
{
    return (x_dtype == VQHIP_F32 || x_dtype == VQHIP_BF16) && vqhip_screen_supported(N, D, C) && (D % 4 == 0) && C <= 8192 ? 1 : 0;
}

extern "C" int vqhip_vq_train_step(const vqhip_vq_step_t *s, void *stream)
{
    if (!s) VQ_FAIL(VQHIP_EINVAL, "vq_train_step: null argument block");
    const int64_t N = s->N;
    const int D = (int)s->D, C = (int)s->C, x_dtype = (int)s->x_dtype;
    if (!s->x || !s->embed || !s->idx_out || !s->stats || !s->packed || !s->workspace) VQ_FAIL(VQHIP_EINVAL, "vq_train_step: null pointer");
    if (s->fold && (!s->embed_avg || !s->cluster_size)) VQ_FAIL(VQHIP_EINVAL, "vq_train_step: fold needs embed_avg and cluster_size");
    if (s->metric != VQHIP_EUCLID && s->metric != VQHIP_COSINE_PRENORM)
        VQ_FAIL(VQHIP_EINVAL, "vq_train_step: metric %d (VQHIP_EUCLID, or VQHIP_COSINE_PRENORM on rows normalised by vqhip_l2norm_rows)", (int)s->metric);
    const int metric = (int)s->metric;
    if (!vqhip_vq_step_supported(x_dtype, N, D, C)) VQ_FAIL(VQHIP_EDIM, "vq_train_step: N=%lld D=%d C=%d dtype=%d outside the fused step", (long long)N, D, C, x_dtype);
    if (s->workspace_bytes < vqhip_vq_step_workspace_bytes(N, C)) VQ_FAIL(VQHIP_EINVAL, "vq_train_step: workspace too small");
    if ((((uintptr_t)s->workspace) & 255) || (((uintptr_t)s->stats) & 15) || (((uintptr_t)s->packed) & 15))
        VQ_FAIL(VQHIP_EALIGN, "vq_train_step: workspace must be 256-byte aligned, stats / packed 16-byte aligned");
    // Row chunks (s->chunks > 1): the search of chunk k + 1 runs on `stream` while the statistics of chunk k (histogram, scan,
    // scatter, segmented sum + loss -- HBM-bound, ~0.18 ms of a 0.84 ms cfg-2 step) run on s->side_stream; counts and sums are
    // order-free (integer counts, fp32 atomics), so only the LAST chunk's statistics -- issued on `stream` behind the side stream's
    // -- and the fold remain behind the search.  Needs the caller's side stream and K events; without them: one chunk.
    int K = (int)s->chunks;
    if (K < 1 || !s->side_stream) K = 1;
    if (K > VQ_STEP_MAX_CHUNKS) K = VQ_STEP_MAX_CHUNKS;
    const int64_t rpc = vqhip_vq_step_chunk_rows(N, K);
    K = (int)((N + rpc - 1) / rpc);
    for (int k = 0; k < K && K > 1; ++k)
        if (!s->events[k]) VQ_FAIL(VQHIP_EINVAL, "vq_train_step: %d chunks need events[0..%d]", K, K - 1);
    hipStream_t st = (hipStream_t)stream, side = (hipStream_t)s->side_stream;
    const int es = (x_dtype == VQHIP_BF16) ? 2 : 4;

    char *ws = (char *)s->workspace;
    void *ws_screen[VQ_STEP_MAX_CHUNKS], *ws_stats[VQ_STEP_MAX_CHUNKS];
    int64_t rows0[VQ_STEP_MAX_CHUNKS], nrows[VQ_STEP_MAX_CHUNKS], part0[VQ_STEP_MAX_CHUNKS];
    int64_t n_part = 0;
    for (int k = 0; k < K; ++k) {
        rows0[k] = (int64_t)k * rpc;
        nrows[k] = N - rows0[k] < rpc ? N - rows0[k] : rpc;
        ws_screen[k] = ws;                         ws += step_ws_screen(nrows[k]);
        ws_stats[k] = ws;                          ws += align_up(vqhip_ema_workspace_bytes(nrows[k], C), 256);
        part0[k] = n_part;                         n_part += seg_work_items(nrows[k], C);
    }
    float *denom = (float *)ws;                    ws += align_up((size_t)C * 4, 256);
    double *partials = (double *)ws;
    float *embed_sum = s->stats, *count = s->stats + (size_t)C * D;

    // VQHIP_STEP_FOLD=1: embed_avg / embed / the loss folded by the segmented sum's own waves (SegArgs FOLD) instead of the launch of
    // vq_step_fold_kernel behind it.  Built and measured in round 6: 0.8414 vs 0.8409 ms per cfg-2 step -- the tickets, the last waves'
    // folds and the last workgroup's reduction cost the kernel's tail what the launch cost -- so it is not the default.
    static int fold_env = -1;
    if (fold_env < 0) { const char *e = getenv("VQHIP_STEP_FOLD"); fold_env = (e && e[0] == '1') ? 1 : 0; }
    const bool fold_in_sum = fold_env && D <= 512;
    ZeroArgs z;
    for (int r = 0; r < VQ_ZERO_REGIONS; ++r) { z.p[r] = nullptr; z.n[r] = 0; }
    z.p[0] = (unsigned *)s->stats;                                             z.n[0] = (unsigned)((size_t)C * D + C);
    for (int k = 0; k < K; ++k) {
        z.p[1 + 2 * k] = (unsigned *)ws_screen[k];                             z.n[1 + 2 * k] = 4;              // list header
        z.p[2 + 2 * k] = (unsigned *)ws_stats[k];                              z.n[2 + 2 * k] = (unsigned)C;    // the histogram
    }
    if (int rc = pack_codebook_impl(s->embed, C, D, s->packed, stream, 1, &z)) return rc;      // (the zeroing rides in the pack kernel's grid)
    const void *qsrc = (x_dtype == VQHIP_BF16) ? (const void *)((const char *)s->packed + packed_bf16_offset(C, D)) : (const void *)s->embed;
    if (s->ev_search_begin) (void)hipEventRecord((hipEvent_t)s->ev_search_begin, st);
    for (int k = 0; k < K; ++k) {
        const char *xk = (const char *)s->x + rows0[k] * s->ldx * es;
        char *qk = s->q_out ? (char *)s->q_out + rows0[k] * s->ldq * es : nullptr;
        if (int rc = vq_assign_screened_impl(xk, x_dtype, nrows[k], D, s->ldx, s->packed, s->embed, C, metric, s->idx_out + rows0[k], qk, s->ldq,
                                             nullptr, D, nullptr, nullptr, ws_screen[k], step_ws_screen(nrows[k]), nullptr, nullptr, 1, stream)) return rc;
        if (k + 1 == K && s->ev_search_end) (void)hipEventRecord((hipEvent_t)s->ev_search_end, st);
        // statistics of this chunk: on the side stream behind its search (chunks 0 .. K - 2), on `stream` behind the side stream's
        // work for the last one -- whose scan kernel, the one workgroup that then holds every code's total count, also folds the counts
        // into cluster_size and forms update_ema's denominators
        const bool last = k + 1 == K;
        hipStream_t sk = last ? st : side;
        if (!last) {
            hipError_t e = hipEventRecord((hipEvent_t)s->events[k], st);
            if (e == hipSuccess) e = hipStreamWaitEvent(side, (hipEvent_t)s->events[k], 0);
            if (e != hipSuccess) VQ_FAIL((int)e, "vq_train_step: event record / wait: %s", hipGetErrorString(e));
        } else if (K > 1) {
            hipError_t e = hipEventRecord((hipEvent_t)s->events[K - 1], side);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, (hipEvent_t)s->events[K - 1], 0);
            if (e != hipSuccess) VQ_FAIL((int)e, "vq_train_step: event record / wait: %s", hipGetErrorString(e));
        }
        StatsFuse f;
        f.hist_zeroed = 1;
        f.cs = (s->fold && last) ? s->cluster_size : nullptr;
        f.denom = (s->fold && last) ? denom : nullptr;
        f.omd = (float)s->one_minus_decay;
        f.eps = (float)s->eps;
        if (s->fold && last && fold_in_sum) {      // embed_avg / embed / the loss folded by the segmented sum's own waves
            f.fold_embed_avg = s->embed_avg; f.fold_embed = s->embed;
            f.fold_cosine = metric != VQHIP_EUCLID ? 1 : 0;
            f.loss_partials = partials; f.n_loss = n_part; f.loss_scale = s->loss_scale; f.loss_out = s->loss_out;
        }
        if (int rc = ema_accumulate_impl(xk, x_dtype, nrows[k], D, s->ldx, s->idx_out + rows0[k], 1, nullptr, VQHIP_EUCLID,
                                         s->row_mask ? s->row_mask + rows0[k] : nullptr, C, count, embed_sum, ws_stats[k],
                                         vqhip_ema_workspace_bytes(nrows[k], C), qsrc, partials + part0[k], (void *)sk, &f)) return rc;
    }
    if (s->fold && fold_in_sum) return 0;
    if (s->fold) {
        hipLaunchKernelGGL(vq_step_fold_kernel, dim3((unsigned)((C + 3) / 4 + 1)), dim3(256), 0, st, s->embed_avg, s->embed, embed_sum, denom,
                           C, D, (float)s->one_minus_decay, metric != VQHIP_EUCLID ? 1 : 0, partials, n_part, s->loss_scale, s->loss_out);
        return launch_status("vq_step_fold_kernel");
    }
    if (s->loss_out) {
        hipLaunchKernelGGL(vq_reduce_kernel, dim3(1), dim3(256), 0, st, partials, n_part, s->loss_scale, s->loss_out);
        return launch_status("vq_reduce_kernel");
    }
    return 0;
}
