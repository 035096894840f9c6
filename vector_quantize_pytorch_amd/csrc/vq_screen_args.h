// vq_screen_args.h -- argument block and vector types shared by the screening kernels (vq_screen.hip, vq_screen_c.hip).
#pragma once

#include "vqhip_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef VQS_WAVES
#define VQS_WAVES 4          // waves per workgroup (2 workgroups of 4 or 1 of 8 per CU: 2 waves per SIMD either way)
#endif

#define VQ_SCREEN_ROWS (VQS_WAVES * 64)   // rows per workgroup: waves x 2 row blocks x 32
#define VQ_SEG_MAX 512                    // list segments of the persistent screening kernel (= its largest grid: 2 workgroups x 256 CUs)

struct ScreenArgs {
    const void *x;                     // rows, bf16 or fp32
    int64_t N;
    int64_t ldx;
    const char *tiles16;               // fp16 screening tiles (vq_screen16_kernel)
    int n_tiles16;                     // fp16 tiles incl. padding tiles (multiple of VQ_F16_TILE_GROUP)
    const unsigned short *embed_bf16;  // bf16 codebook copy inside the packed codebook
    const float *embed;                // fp32 codebook (q rows of fp32 I/O)
    const unsigned *scalars;           // [0] = float bits of max ||c||^2, [1] = float bits of max ||c - c_f16||, [2] = sc
    int C;
    int64_t *idx_out;
    void *q_out;                       // nullable, x's dtype
    int64_t ldq;
    void *resid_out;                   // nullable, x's dtype: x - q (the next residual-VQ stage's input, rvq.py:524)
    int64_t ldr;
    double *sqerr_partial;             // nullable, one entry per workgroup
    const uint8_t *row_mask;
    int *flag_count;                   // [0] open rows (list front), [1] pair rows (list back)
    int *flag_rows;
    unsigned long long *flag_keys;     // [N] keys of the exact pass, preset to ~0 for every appended row
    float *dbg;                        // nullable [N, 4]: t_best, t_second, eps_t, flagged
    // residual chain (vq_screen16_kernel, fp32 rows): this stage's rows are x - prev_embed[prev_idx], formed in the prologue from
    // the PREVIOUS stage's input and indices and written to x_out (the exact passes and the statistics read them there)
    int64_t idx_stride;                // idx_out[row * idx_stride] (a column of an [N, Q] index tensor)
    const int64_t *prev_idx;           // nullable
    int64_t prev_idx_stride;
    const float *prev_embed;           // [C_prev, D] fp32
    float *x_out;
    int64_t ldxo;
    // start offset (vq_screen16_kernel): the second workgroup of every CU in the launch's FIRST round (blockIdx.x < 2 x CUs) spins for
    // stagger x 1024 cycles, so that the two workgroups of a CU alternate between their memory phase (rows, previous codes, x_out) and
    // their sweep instead of meeting the whole chip in both
    int stagger;
    int stagger_first;
    // segmented lists (vq_screenc_kernel: every workgroup appends to its own segment, no global atomics; vq_compact_lists_kernel
    // packs the segments into flag_rows / flag_keys and writes flag_count)
    // several heads in one launch (vqhip_assign_screened_batched: blockIdx.y = head): byte strides between consecutive heads' rows,
    // packed codebooks, fp32 codebooks, index / q outputs and workspaces.  heads <= 1: a plain launch.  (No residual / squared-error
    // outputs in a batched launch.)  Round 6: the residual chain batches too -- the G groups of GroupedResidualVQ (rvq.py:634-724)
    // are the heads: prev_idx sits in the same index block as idx_out (hs_idx), prev_embed is laid out like embed (hs_embed), x_out
    // has its own stride (hs_xo: the stage inputs are [G, N, D] blocks, the caller's rows may be feature chunks of a wider tensor).
    int heads;
    int64_t hs_x, hs_packed, hs_embed, hs_idx, hs_q, hs_ws, hs_xo;
    int *seg_counts;                   // [2 * VQ_SEG_MAX]: open, pair entries per segment
    int *seg_rows;                     // [VQ_SEG_MAX * seg_cap]
    unsigned long long *seg_keys;      // [VQ_SEG_MAX * seg_cap]
    int seg_cap;                       // entries per segment: the rows its workgroup handles
#ifdef VQ_TRACE
    long long *trace;
#endif
};


// the argument block of head blockIdx.y of a batched launch
__device__ __forceinline__ ScreenArgs vq_head_screen_args(const ScreenArgs &a0)
{
    if (a0.heads <= 1) return a0;
    ScreenArgs a = a0;
    const int64_t h = blockIdx.y;
    a.x = (const char *)a0.x + h * a0.hs_x;
    a.tiles16 = a0.tiles16 + h * a0.hs_packed;
    a.embed_bf16 = (const unsigned short *)((const char *)a0.embed_bf16 + h * a0.hs_packed);
    a.scalars = (const unsigned *)((const char *)a0.scalars + h * a0.hs_packed);
    a.embed = (const float *)((const char *)a0.embed + h * a0.hs_embed);
    a.idx_out = (int64_t *)((char *)a0.idx_out + h * a0.hs_idx);
    if (a0.q_out) a.q_out = (char *)a0.q_out + h * a0.hs_q;
    a.flag_count = (int *)((char *)a0.flag_count + h * a0.hs_ws);
    a.flag_rows = (int *)((char *)a0.flag_rows + h * a0.hs_ws);
    a.flag_keys = (unsigned long long *)((char *)a0.flag_keys + h * a0.hs_ws);
    if (a0.prev_idx) {
        a.prev_idx = (const int64_t *)((const char *)a0.prev_idx + h * a0.hs_idx);
        a.prev_embed = (const float *)((const char *)a0.prev_embed + h * a0.hs_embed);
        if (a0.x_out) a.x_out = (float *)((char *)a0.x_out + h * a0.hs_xo);
    }
    return a;
}

// persistent screening kernel with the cyclic tile stream (vq_screen_c.hip): 1 if it serves this launch, and the launch itself
int vq_screenc_eligible(const ScreenArgs &a, int x_dtype, int DT);
int vq_screenc_launch(const ScreenArgs &a, int metric_is_cosine, hipStream_t st);
