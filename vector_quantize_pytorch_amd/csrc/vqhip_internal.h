// vqhip_internal.h -- declarations shared by the translation units of libvqhip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vqhip.h"

extern thread_local char vq_g_err[256];
#define VQ_FAIL(code, ...)                                 \
    do {                                                   \
        snprintf(vq_g_err, sizeof vq_g_err, __VA_ARGS__);  \
        return (code);                                     \
    } while (0)

int vq_launch_status(const char *what);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: remember it per (kernel instantiation, device),
// so that a process driving several GPUs raises the 64-KiB default on each of them (bit d = done on device d).
struct VqAttrOnce { unsigned long long mask = 0; };
static inline int vq_set_max_smem(VqAttrOnce &once, const void *fn, int bytes, const char *what)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= 0 && dev < 64 && ((once.mask >> dev) & 1ull)) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) VQ_FAIL((int)e, "hipFuncSetAttribute(%s, %d bytes of LDS): %s", what, bytes, hipGetErrorString(e));
    if (dev >= 0 && dev < 64) once.mask |= 1ull << dev;   // idempotent: a race between threads only repeats the call
    return 0;
}

__device__ __forceinline__ float vq_bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short vq_f32_to_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// bf16 tensor arithmetic of the reference (x - q on bf16 tensors: fp32 subtract, RNE back to bf16), 4 packed elements
__device__ __forceinline__ uint2 vq_bf16x4_sub(uint2 x, uint2 g)
{
    const float d0 = __uint_as_float(x.x << 16) - __uint_as_float(g.x << 16);
    const float d1 = __uint_as_float(x.x & 0xffff0000u) - __uint_as_float(g.x & 0xffff0000u);
    const float d2 = __uint_as_float(x.y << 16) - __uint_as_float(g.y << 16);
    const float d3 = __uint_as_float(x.y & 0xffff0000u) - __uint_as_float(g.y & 0xffff0000u);
    uint2 r;
    r.x = (unsigned)vq_f32_to_bf16_rne(d0) | ((unsigned)vq_f32_to_bf16_rne(d1) << 16);
    r.y = (unsigned)vq_f32_to_bf16_rne(d2) | ((unsigned)vq_f32_to_bf16_rne(d3) << 16);
    return r;
}

// feature-tile width: D is padded to DT in the packed codebook
static inline int vq_pick_dt(int D)
{
    if (D <= 32) return 32;
    if (D <= 64) return 64;
    if (D <= 128) return 128;
    if (D <= 256) return 256;
    if (D <= 512) return 512;
    return 0;
}

// ---- layout of the packed codebook (vqhip_pack_codebook), in bytes from its start ----------------
//   [0)                     fp32 A-operand tiles of the exact kernel: tiles * (128*DT + 1024)
//   [+4096)                 tail pad (the staged tile copy over-reads <= 3 KiB)
//   [bf16_offset)           codebook rounded to bf16, [C, D] row-major (q / loss of bf16 I/O), 16-byte padded
//   [scalars_offset)        64 bytes: [0] float bits of max_c ||c||^2, [1] float bits of rho, [3] of r0: ||c - c_f16|| <= rho ||c|| + r0
//                           for every code (round 6; was the codebook-wide max_c ||c - c_f16||), [2] int sc: the fp16 tiles hold
//                           c * 2^sc, [4] ~(float bits of min_c ||c||^2), rest reserved
//   [f16_offset)            fp16 A-operand tiles of the single-pass screening kernel (vq_screen16_kernel):
//                           tiles16 * (64*DT + 1024), tiles16 = tiles rounded up to a multiple of VQ_F16_TILE_GROUP
//                           (padding tiles score -3e38), then an 8192-byte tail pad.  Tile tail (1 KiB of floats): [0, 32) the
//                           accumulator's start value -||c||^2/2 (+ the code's row-independent error share), [32, 64) an upper
//                           bound of ||c|| (0 for padding codes)
#define VQ_F16_TILE_GROUP 8
#define VQ_PACKED_SCALARS_BYTES 64
__host__ __device__ static inline size_t vq_tile_bytes(int DT) { return (size_t)128 * DT + 1024; }
__host__ __device__ static inline size_t vq_tile16_bytes(int DT) { return (size_t)64 * DT + 1024; }
static inline size_t vq_tiles16(int C)
{
    const size_t tiles = ((size_t)C + 31) / 32;
    return (tiles + VQ_F16_TILE_GROUP - 1) / VQ_F16_TILE_GROUP * VQ_F16_TILE_GROUP;
}
static inline size_t vq_packed_bf16_offset(int C, int D)
{
    if (D > 512) return ((size_t)C * 4 + 255) / 256 * 256;      // wide dims (vq_wide.hip): y2 [C] floats, 256-byte padded, then the bf16 copy
    const size_t tiles = ((size_t)C + 31) / 32;
    return tiles * vq_tile_bytes(vq_pick_dt(D)) + 4096;
}
static inline size_t vq_packed_scalars_offset(int C, int D)
{
    return vq_packed_bf16_offset(C, D) + (((size_t)C * D * 2 + 15) & ~(size_t)15) + 8192;   // (+ pad: row-cooperative over-reads)
}
static inline size_t vq_packed_f16_offset(int C, int D) { return vq_packed_scalars_offset(C, D) + VQ_PACKED_SCALARS_BYTES; }
static inline size_t vq_packed_total_bytes(int C, int D)
{
    return vq_packed_f16_offset(C, D) + vq_tiles16(C) * vq_tile16_bytes(vq_pick_dt(D)) + 8192;
}

// exact fp32-MFMA assignment (vqhip.hip) restricted to the rows listed in row_list[0 .. *row_count), both on the
// device; x and q share a dtype (fp32 / bf16) with D == DT and vector-aligned rows; metric VQHIP_EUCLID or
// VQHIP_COSINE_PRENORM.  keys: N u64, entries [0 .. row_count[0]) preset to ~0 by the list builder.  with_pairs: rows whose
// winner is one of two known codes sit at list positions N - 1 - p, p < row_count[1], their keys hold the candidates
// (c1 | c2 << 32); vq_pair_kernel decides them with two exact distances instead of a codebook sweep.
// sqerr_partial (nullable) receives VQ_FINISH_BLOCKS entries.
#define VQ_FINISH_BLOCKS 512
// several heads in one launch (blockIdx.y = head): byte strides between consecutive heads' rows, packed codebooks, fp32 codebooks,
// the gather source of the finish kernel (bf16 copy inside the packed buffer, or embed), index / q outputs and workspaces
struct VqHeadStrides {
    int heads;          // <= 1: a plain launch
    int64_t x, packed, embed, codes, idx, q, ws;
    int64_t xo = 0;     // a chained stage's x_out (the exact passes read their rows there)
};
int vq_assign_listed(const void *x, int x_dtype, int metric, int64_t N, int D, int64_t ldx, const float *packed, const float *embed, int C,
                     int64_t *idx_out, int64_t idx_stride, void *q_out, int64_t ldq, void *resid_out, int64_t ldr, double *sqerr_partial,
                     const uint8_t *row_mask, const int *row_list, const int *row_count, unsigned long long *keys, int with_pairs,
                     hipStream_t st, const VqHeadStrides *hs = nullptr);

// arrival counters of the merged exact-pass launch: one per 128-row chunk of the open-row list (an even number of ints: the row list
// behind them stays 8-byte aligned)
static inline size_t vq_screen_done_ints(int64_t N) { return (size_t)((((N > 0 ? N : 1) + 127) / 128 + 1) & ~(int64_t)1); }
// The listed exact passes of a residual-chain stage as ONE launch (vq_tail_kernel, vqhip.hip): workgroups [0, gx) sweep the open rows,
// the rest decide the pair rows; both write idx_out themselves (a split sweep: the workgroup that arrives last at a chunk's counter in
// `done` -- zeroed by the caller -- reads the chunk's keys back).  Index output only.  Measured slower than the three separate launches
// (vqhip.hip, vq_tail_enabled): only with VQHIP_TAIL=1.
int vq_tail_enabled();
int vq_assign_listed_direct(const void *x, int x_dtype, int metric, int64_t N, int D, int64_t ldx, const float *packed, const float *embed, int C,
                            int64_t *idx_out, int64_t idx_stride, const int *row_list, const int *row_count, unsigned long long *keys, int *done,
                            hipStream_t st, const VqHeadStrides *hs = nullptr);

// the screened assignment behind vqhip_assign_screened / vqhip_assign_screened_chain (vq_screen.hip); header_zeroed says that the
// caller (the fused train step, vqhip_vq_train_step) has already zeroed the workspace's 16-byte list header on this stream
int vq_assign_screened_impl(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed,
                            const float *embed, int C, int metric, int64_t *idx_out, void *q_out, int64_t ldq,
                            void *resid_out, int64_t ldr, double *sqerr_partial, const uint8_t *row_mask,
                            void *workspace, size_t workspace_bytes, float *debug_out, const vqhip_chain_t *chain,
                            int header_zeroed, void *stream, const VqHeadStrides *hs = nullptr);

// statistics of H (rows, codebook) pairs in one launch set, every head stride in bytes (vqhip.hip); qsrc: the loss' code rows in x's
// dtype (nullable with sqerr_partial)
int vq_ema_accumulate_heads(const void *x, int x_dtype, int H, int64_t N, int D, int64_t ldx, int64_t hs_x, const int64_t *idx,
                            int64_t idx_stride, int64_t hs_idx, const uint8_t *row_mask, int C, float *count, float *embed_sum,
                            int64_t hs_stats, void *workspace, int64_t hs_ws, int hist_zeroed, const void *qsrc, int64_t hs_qsrc,
                            double *sqerr_partial, int64_t hs_sq, void *stream);

// ---- codebook dims 512 < D <= 2048 (vq_wide.hip): plain exact kernels behind the same entry points --------------------------------
#define VQ_WIDE_MAX_D 2048
static inline bool vq_is_wide(int D) { return D > 512 && D <= VQ_WIDE_MAX_D; }
size_t vq_wide_packed_bytes(int C, int D);
int vq_wide_pack(const float *embed, int C, int D, float *packed, int H, void *stream);
int vq_wide_row_sumsq(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, float *out, void *stream);
int vq_wide_l2norm_rows(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, void *out, int64_t ldo, void *stream);
int vq_wide_assign(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed, const float *embed, int C, int metric,
                   int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq, float *best_out, float *rnorm_out, double *sqerr_partial,
                   const uint8_t *row_mask, void *stream);
int vq_wide_ema_embed(float *embed_avg, float *embed, const float *embed_sum, const float *weight, const float *denom, int H, int C, int D,
                      float omd, int cosine, int do_lerp, int do_update, int64_t hs_sum, void *stream);
int vq_wide_decode_sum(const int64_t *idx, int64_t N, int Q, const float *embed, int64_t qstride, int C, int D, void *out, int out_dtype,
                       int64_t ldo, void *stream);
