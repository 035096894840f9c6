/*
 * vqhip.h -- C ABI of libvqhip.so, the MI355X (gfx950 / CDNA4) vector-quantization hot path.
 *
 * The reference (lucidrains/vector-quantize-pytorch v1.31.0) is pure Python and has no FFI; its
 * seam is `VectorQuantize._codebook(x, ...) -> (quantize, embed_ind, dist)`
 * (vector_quantize_pytorch.py:1176, class Codebook :349-791).  Each entry point below replaces the
 * ATen op sequence of one stretch of that class; the reference lines are cited per function
 * ("vqp.py" = vector_quantize_pytorch/vector_quantize_pytorch.py, "rvq.py" = residual_vq.py).
 *
 * Conventions
 *  - plain pointers + sizes, no framework types.  Every pointer is a DEVICE pointer.
 *  - the caller owns every buffer, including workspaces; the library never allocates, frees or
 *    synchronises.  All work is enqueued on `stream` (a hipStream_t passed as void*).
 *  - return value: 0 on success, a negative VQHIP_E* code on argument errors, or a positive
 *    hipError_t from the launch.  vqhip_last_error() returns a thread-local message.
 *  - dtype codes: VQHIP_F32 = 0, VQHIP_BF16 = 1.  metric: VQHIP_EUCLID / VQHIP_COSINE / VQHIP_COSINE_PRENORM.
 *  - supported shapes: 1 <= D <= 2048, C >= 1, N >= 0; rows addressed as base + n * ld (elements).  D <= 512: the tuned kernels
 *    (rows resident in VGPRs as MFMA operands).  512 < D <= 2048 (round 5, csrc/vq_wide.hip): plain exact kernels behind
 *    vqhip_pack_codebook / vqhip_assign / vqhip_row_sumsq / vqhip_l2norm_rows / vqhip_ema_accumulate* (D % 4 == 0) / vqhip_ema_finalize /
 *    vqhip_decode_sum / vqhip_route_* / vqhip_expire_* -- same arithmetic contract (ATen-order norms incl. the cascade level of rows
 *    longer than 512, one ascending fp32 FMA chain per dot product); the screened search, the fused step / residual loop, score rows,
 *    top-k and batched heads return VQHIP_EDIM there.
 */
#ifndef VQHIP_H
#define VQHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQHIP_F32  0
#define VQHIP_BF16 1

#define VQHIP_EUCLID 0
#define VQHIP_COSINE 1          /* l2-normalise the rows in the kernel prologue, then max dot product */
#define VQHIP_COSINE_PRENORM 2  /* rows are already unit-norm: max dot product only                    */

#define VQHIP_EINVAL   (-1)   /* bad argument (null pointer, negative size, unknown dtype) */
#define VQHIP_EDIM     (-2)   /* D outside the supported range                              */
#define VQHIP_EALIGN   (-3)   /* pointer / stride alignment requirement violated            */

#define VQHIP_ASSIGN_ROWS_PER_BLOCK 128  /* rows one workgroup of vqhip_assign owns */

const char *vqhip_version(void);
const char *vqhip_last_error(void);

/* ---- codebook preparation -------------------------------------------------------------------
 * Re-tiles the fp32 codebook `embed` [C, D] into the MFMA A-operand order the assign kernel
 * streams through LDS, and appends ||c||^2 per code summed in ATen's CPU order (the `y2` term
 * of cdist, vqp.py:60).  Must be re-run whenever `embed` changes.
 * packed: caller buffer of vqhip_packed_bytes(C, D) bytes, 16-byte aligned. */
size_t vqhip_packed_bytes(int C, int D);
int vqhip_pack_codebook(const float *embed, int C, int D, float *packed, void *stream);

/* ---- nearest-code assignment + gather + commitment-loss partials ------------------------------
 * Replaces: cdist (vqp.py:58-62) / cosine einsum (:741), the negate + argmax of gumbel_sample's
 * deterministic branch (:134-145), the one-hot gather (:766, :779-781), l2norm of the input for
 * the cosine metric (:37-38 applied at :1159) and the squared-error sum of F.mse_loss (:1327).
 *
 *  x          [N, D] rows at stride ldx (elements), dtype x_dtype.
 *  packed     output of vqhip_pack_codebook for the SAME embed.
 *  embed      [C, D] fp32 (rows are copied verbatim into q_out).
 *  idx_out    [N] int64: index of the first code attaining the minimum distance (maximum
 *             similarity for cosine) -- ties resolve to the lowest index like ATen argmax.
 *  q_out      nullable; [N, D] rows at stride ldq, dtype q_dtype: embed[idx] (rounded RNE for bf16).
 *  best_out   nullable; [N] fp32: the winning distance (euclid) or similarity (cosine).
 *  rnorm_out  nullable; [N] fp32: euclid -> ||x||^2 in ATen order; cosine -> max(||x||, 1e-6).
 *  sqerr_partial  nullable; [vqhip_assign_blocks(N)] doubles: per-workgroup sum over its rows of
 *             sum_d (q - x)^2 (x = l2-normalised x for cosine), rows with row_mask == 0 excluded.
 *  row_mask   nullable; [N] bytes (vqp.py:599-600, 1317-1325 semantics: 0 = padding row).
 */
int64_t vqhip_assign_blocks(int64_t N);
int vqhip_assign(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                 const float *packed, const float *embed, int C, int metric,
                 int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq,
                 float *best_out, float *rnorm_out, double *sqerr_partial,
                 const uint8_t *row_mask, void *stream);

/* ---- l2norm of rows ------------------------------------------------------------------------------
 * Replaces l2norm (vqp.py:37-38) as applied to the input at :1159: out = x / max(||x||, 1e-6) with ||x||^2 summed in
 * ATen's CPU order and, for bf16 tensors, norm and quotient rounded to bf16 as the reference's bf16 ops do -- the same
 * arithmetic vqhip_assign(metric VQHIP_COSINE) applies internally.  D in {32, 64, 128, 256, 512}; rows aligned to 4 elements. */
int vqhip_l2norm_rows(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, void *out, int64_t ldo, void *stream);
/* Its backward, the gradient autograd derives for F.normalize (the reference keeps the l2norm of :1159 in the graph when the input
 * requires grad): out = g / n - [||x|| >= eps] x / ||x|| * sum_d(g_d x_d) / n^2 with n = max(||x||, eps), one pass over x and g instead
 * of the quotient's, the clamp's and the norm's backward kernels.  x, g, out in x_dtype; D % 4 == 0, D <= 512; rows aligned to 4 elements. */
int vqhip_l2norm_rows_bwd(const void *x, const void *g, int x_dtype, int64_t N, int D, int64_t ldx, int64_t ldg,
                          void *out, int64_t ldo, void *stream);

/* ---- screened assignment (D in {32, 64, 128, 256, 512}) -------------------------------------------------
 * metric: VQHIP_EUCLID, or VQHIP_COSINE_PRENORM on rows already normalised (vqhip_l2norm_rows).
 * Same contract as vqhip_assign(same metric, q in x's dtype): idx_out bit-identical to the exact kernel, q_out
 * the gathered code rows, sqerr_partial the squared-error partials -- but the codebook sweep runs on the fp16 MFMA pipe
 * against ONE fp16 copy of the codebook made by vqhip_pack_codebook (rows: bf16 values scaled by a power of two are exact
 * fp16 operands; fp32 rows are rounded to one fp16 operand set and the measured residual is charged), tracking the three
 * best scores per row.  A row whose best-vs-second margin exceeds the proven error bound (csrc/vq_screen.hip) is final;
 * a row with two candidates inside the bound is decided by two exact distances; the rest is re-evaluated by the exact
 * fp32-MFMA sweep -- all on the same stream, before the call's work completes.  Replaces the same reference lines as
 * vqhip_assign.
 *   supported:      vqhip_screen_supported(N, D, C) != 0; x rows 16-byte aligned, q rows aligned to 4 elements
 *   workspace:      vqhip_screen_workspace_bytes(N) bytes, 8-byte aligned; on completion ((int *)workspace)[0] is the
 *                   number of rows that took the exact sweep, [1] the number decided between two candidates (diagnostic)
 *   sqerr_partial:  nullable, vqhip_screen_partials(N, x_dtype) doubles, all written; feed them to vqhip_reduce_partials
 *   resid_out:      nullable [N, ldr] in x's dtype: x - q in the reference's tensor arithmetic (bf16 tensors subtract in
 *                   fp32 and round to bf16), i.e. the input of the next ResidualVQ stage (rvq.py:524); q_out may be null
 *   debug_out:      nullable [N, 4] floats: best score, runner-up, certification threshold, 1.0 if re-evaluated */
int vqhip_screen_supported(int64_t N, int D, int C);
size_t vqhip_screen_workspace_bytes(int64_t N);
int64_t vqhip_screen_blocks(int64_t N, int x_dtype);
int64_t vqhip_screen_partials(int64_t N, int x_dtype);
int vqhip_assign_screened(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed,
                          const float *embed, int C, int metric, int64_t *idx_out, void *q_out, int64_t ldq,
                          void *resid_out, int64_t ldr, double *sqerr_partial, const uint8_t *row_mask,
                          void *workspace, size_t workspace_bytes, float *debug_out, void *stream);

/* H searches in one set of launches (grid dimension y = head): the heads of a multi-head VectorQuantize with separate codebooks
 * (vqp.py:1044-1049) or of RandomProjectionQuantizer (random_projection_quantizer.py:37-59), which the reference runs as one batched
 * einsum over the head axis.  Head h's buffers sit h strides behind head 0's: x at x_hstride elements; packed at
 * vqhip_packed_bytes(C, D) bytes (made by vqhip_pack_codebook_batched from embed [H, C, D]); embed at C * D floats; idx_out [H, N];
 * q_out (nullable) at q_hstride elements; workspace at vqhip_screen_batched_ws_stride(N) bytes (H of them).  row_mask (nullable, [N])
 * is shared by the heads.  Index and q outputs only; contract per head as vqhip_assign_screened. */
int vqhip_pack_codebook_batched(const float *embed, int H, int C, int D, float *packed, void *stream);
size_t vqhip_screen_batched_ws_stride(int64_t N);
int vqhip_assign_screened_batched(const void *x, int x_dtype, int H, int64_t N, int D, int64_t ldx, int64_t x_hstride,
                                  const float *packed, const float *embed, int C, int metric, int64_t *idx_out,
                                  void *q_out, int64_t ldq, int64_t q_hstride, const uint8_t *row_mask,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* The exact search of H heads in one launch (grid dimension y = head), for the dims the screened search does not take -- e.g. the
 * 16-wide heads of RandomProjectionQuantizer.  Buffers laid out as for vqhip_assign_screened_batched; rnorm_out (nullable as in
 * vqhip_assign) is [H, N].  Index, q and rnorm outputs only. */
int vqhip_assign_batched(const void *x, int x_dtype, int H, int64_t N, int D, int64_t ldx, int64_t x_hstride,
                         const float *packed, const float *embed, int C, int metric,
                         int64_t *idx_out, void *q_out, int q_dtype, int64_t ldq, int64_t q_hstride,
                         float *rnorm_out, const uint8_t *row_mask, void *stream);

/* Residual chain for the stages of a residual VQ (reference: the loop body of ResidualVQ.forward, residual_vq.py:469-568, with
 * `residual = residual - quantized.detach()` at :524).  A stage's screening kernel forms its own input in its prologue,
 * x - prev_embed[prev_idx] in fp32 (exactly the x - q the previous stage would have written), from the PREVIOUS stage's input x
 * and indices, and stores it to x_out, where the exact passes of this stage (and the caller's statistics pass) read it -- so no
 * stage re-reads its input to write a residual.  prev_idx == NULL: plain search of x (first stage); idx_stride lets every stage
 * write its column of an [N, Q] index tensor.  A chained stage (prev_idx given): fp32 rows, D in {32, 64, 128, 256}
 * (vqhip_screen_chain_supported); with prev_idx == NULL any rows vqhip_assign_screened takes (bf16, D = 512: the stages of a loop
 * whose inputs vqhip_route_residual wrote).  Euclidean; no q / residual / squared-error outputs (the statistics pass sums the loss:
 * vqhip_ema_accumulate_sqerr). */
typedef struct {
    int64_t idx_stride;            /* idx_out[n * idx_stride] */
    const int64_t *prev_idx;       /* nullable: previous stage's indices, prev_idx[n * prev_idx_stride] */
    int64_t prev_idx_stride;
    const float *prev_embed;       /* previous stage's codebook [C_prev, D] fp32 */
    void *x_out;                   /* [N, D] fp32 at row stride ldxo: receives this stage's input */
    int64_t ldxo;
    int64_t route_mode;            /* must be 0.  (Round 4's first form subtracted the previous layer's ROUTED value here; that is a
                                      kernel of its own now -- vqhip_route_residual -- whose output is passed as x with prev_idx NULL.) */
    int64_t header_zeroed;         /* != 0: the caller has zeroed the first 16 bytes of `workspace` on this stream (no memset launch) */
} vqhip_chain_t;
int vqhip_screen_chain_supported(int x_dtype, int D);
int vqhip_assign_screened_chain(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed,
                                const float *embed, int C, int metric, int64_t *idx_out, const uint8_t *row_mask,
                                void *workspace, size_t workspace_bytes, const vqhip_chain_t *chain, void *stream);

/* ---- the residual loop as one call (round 5; groups: round 6) ---------------------------------------
 * Everything ResidualVQ.forward's loop (residual_vq.py:469-568) launches between packing the codebooks and decoding the sum, issued from C:
 * the Q chained screened searches (vqhip_assign_screened_chain: stage q forms x_prev - code in its prologue, residual_vq.py:524), for a
 * training step whose input requires grad the routed residuals instead (vqhip_route_residual, vqp.py:1225-1233), and -- when `stats` is
 * given -- every stage's EMA statistics + commitment-loss partials (vqhip_ema_accumulate_prezeroed, vqp.py:599-606, 1327).  The rows may
 * be split into `chunks` contiguous row chunks of vqhip_rvq_chain_chunk_rows(N, chunks) rows, each running its own chain on its own
 * stream (chunk 0 on `stream`), so that one chunk's short exact passes run beside another chunk's screening kernel; stage q's statistics
 * are queued on stats_stream once stage q is final in every chunk.  The caller owns streams, events (re-recorded by every call; the
 * library creates nothing) and buffers.  On return `stream` has been joined with the chunk streams; stats_stream has NOT (join it
 * before reading `stats`).  Results are those of the per-stage entry points, bit for bit (indices) / to the rounding of the segmented
 * sums' atomics (statistics).
 *   inputs      [Q - 1, N, D] in x's dtype, contiguous: receives the inputs of stages 1 .. Q - 1 (stage 0's is x)
 *   workspace   Q x K slices of vqhip_rvq_chain_ws_stride(N, chunks) bytes (K = the number of chunks actually formed), 256-byte
 *               aligned; slice (q, k) starts with that search's counters ([0] rows of the exact sweep, [1] rows decided between two codes)
 *   route_mode  0: `quantized` is the code row; 1 / 2: the straight-through / rotation-trick value (codes: the code rows in x's dtype,
 *               [C, D] per stage at codes_qstride elements, 0 = shared)
 *   stats       nullable [Q, stats_stride] floats = embed_sum [C, D] || count [C] per stage, ZEROED by the caller; stats_ws: Q slices
 *               of stats_ws_stride >= vqhip_ema_batched_ws_stride(N, C) bytes whose first C ints the caller zeroed; sqerr_partial
 *               nullable [Q, sqerr_stride >= vqhip_ema_sqerr_partials(N, C)]
 *   events      at least Q x chunks + 1 hipEvent_t when chunks > 1 or stats_stream differs from `stream` */
typedef struct {
    const void *x; int64_t x_dtype; int64_t N; int64_t D; int64_t ldx;
    const float *packed; int64_t packed_qstride;          /* floats between the stages' packed codebooks (0: one shared codebook) */
    const float *embed; int64_t embed_qstride;            /* floats between the stages' codebooks [C, D] (0: shared, residual_vq.py:302-306) */
    int64_t C; int64_t Q;
    int64_t *idx_out;                                     /* [N, Q] */
    void *inputs;
    const uint8_t *row_mask;                              /* nullable [N] */
    void *workspace; size_t workspace_bytes;
    int64_t route_mode; const void *codes; int64_t codes_qstride;
    float *stats; int64_t stats_stride; void *stats_ws; size_t stats_ws_stride; double *sqerr_partial; int64_t sqerr_stride;
    int64_t chunks; void **chunk_streams;                 /* chunks - 1 streams */
    void *stats_stream;                                   /* nullable / == stream: the statistics follow the loop on `stream` */
    void **events; int64_t n_events;
    /* Round 6: G independent residual loops -- the groups of GroupedResidualVQ (residual_vq.py:634-724; the reference runs them one
     * after the other, loop at :706) -- as ONE launch set, blockIdx.y = group.  groups <= 1: one loop, the fields below are ignored.
     * Group g reads the rows x + g * x_gstride (elements; same N, ldx: the feature chunks of one [N, G D] tensor have x_gstride = D),
     * searches packed + g * packed_gstride / embed + g * embed_gstride (floats; stage q another q * packed_qstride / embed_qstride
     * on top), routes through codes + g * codes_gstride (elements), accumulates into stats + g * stats_gstride (floats) with the
     * workspace stats_ws + g * stats_ws_gstride (bytes) and the loss partials sqerr_partial + g * sqerr_gstride (doubles).
     * Fixed layouts: idx_out [G, N, Q]; inputs [Q - 1, G, N, D]; workspace Q x K x G slices (slice ((q K + k) G + g)); row_mask is
     * shared by the groups.  Results per group: those of a call with groups = 1, bit for bit (indices). */
    int64_t groups;
    int64_t x_gstride, packed_gstride, embed_gstride, codes_gstride, stats_gstride;
    size_t stats_ws_gstride;
    int64_t sqerr_gstride;
    /* Round 6: the decode (quantized_out = sum over the stages of the chosen codes, residual_vq.py:525) split around the last stage.
     * decode_out (nullable, fp32 [N, D] rows at decode_ldo elements, group g decode_gstride elements behind group 0): the sum of the
     * stages [0, Q - 1) is formed on stats_stream while stage Q - 1 is searched, stage Q - 1 is added on `stream` behind the loop --
     * the same additions in the same order as vqhip_decode_sum over all stages, so bit-identical.  Requires Q >= 2, fp32 rows, no
     * row_mask (masked rows are re-indexed to -1 only after the loop), a statistics stream, and one more event (Q x chunks + 2). */
    void *decode_out; int64_t decode_ldo; int64_t decode_gstride;
} vqhip_rvq_chain_t;
int64_t vqhip_rvq_chain_chunk_rows(int64_t N, int chunks);
size_t vqhip_rvq_chain_ws_stride(int64_t N, int chunks);
int vqhip_rvq_chain_forward(const vqhip_rvq_chain_t *c, void *stream);

/* ---- dense scores (rare options only) ------------------------------------------------------------
 * Materialises the tensor the reference calls `dist` (vqp.py:741-743): scores_out[n, c] = -cdist(x_n, c) for the
 * Euclidean metric (same rounding sequence as vqhip_assign), x^_n . c for cosine.  Needed by the options that read
 * the whole row: top-k / beam search (vqp.py:137-138), gumbel sampling (:132-133), cross-entropy / diversity losses
 * (:1242-1261, :1287-1292).  idx_out [N] (argmax) must be given; rnorm_out as in vqhip_assign. */
int vqhip_scores(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                 const float *packed, const float *embed, int C, int metric,
                 float *scores_out, int64_t lds, int64_t *idx_out, float *rnorm_out, void *stream);

/* The same sweep with a streaming log-sum-exp epilogue instead of the N x C store: what F.cross_entropy(dist, codes) needs
 * (cross-entropy "commitment" to the chosen codes, vqp.py:1242-1256, and forward(indices=...), vqp.py:1260-1261):
 *   lse_out[n]    = log sum_c exp(dist[n, c])           (online max / sum per lane, merged per row)
 *   tscore_out[n] = dist[n, target[n]]                   (target null: the winner's score; target[n] < 0: 0, the row is ignored;
 *                                                          target[n] >= C: NaN -- F.cross_entropy raises there, nothing silent)
 * dist as in vqhip_scores.  idx_out [N] (argmax) must be given; rnorm_out as in vqhip_assign. */
int vqhip_scores_lse(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                     const float *packed, const float *embed, int C, int metric, const int64_t *target,
                     float *lse_out, float *tscore_out, int64_t *idx_out, float *rnorm_out, void *stream);

/* ---- fused residual VQ loop ---------------------------------------------------------------------
 * Replaces the per-quantizer loop of ResidualVQ.forward (rvq.py:469-568) for the Euclidean metric and a
 * uniform codebook size: Q successive nearest-code searches on the running residual, which stays in
 * registers between stages.  Stage q uses codebook embed + q*embed_qstride / packed + q*packed_qstride
 * (strides in floats; 0 = all stages share one codebook, rvq.py:302-306).
 *  idx_out        [N, Q] int64 (-1 on rows with row_mask == 0)
 *  resid_out      nullable; [N, Q, D] in x's dtype: the input of every stage (what the reference keeps in
 *                 `all_residuals`, rvq.py:489) for the EMA statistics / expiry afterwards
 *  sqerr_partial  nullable; [Q, 4 * vqhip_assign_blocks(N)] doubles: per-wave sums of (quantized_q - residual_q)^2
 * quantized_out = vqhip_decode_sum(idx_out, ...).  Requires D % 32 == 0. */
int vqhip_rvq_forward(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                      const float *packed, int64_t packed_qstride, const float *embed, int64_t embed_qstride,
                      int C, int Q, int64_t *idx_out, void *resid_out, double *sqerr_partial,
                      const uint8_t *row_mask, void *stream);

/* ---- gradient routing through the quantizer ----------------------------------------------------
 * Replaces straight_through (vqp.py:282-283) and rotate_to / efficient_rotation_trick_transform
 * (vqp.py:287-318) and the backward of F.mse_loss(quantize.detach(), x) (vqp.py:1327).
 *  mode 1 = straight-through, mode 2 = rotation trick.  x, q, g_out, out, grad_x: [N, D] rows, one dtype.
 *  vqhip_route_fwd : out = x + (q - x)                      (mode 1)
 *                    out = |q|/|x| (x - 2 (x.w) w + 2 (x.u) qh)   (mode 2; u, qh, w as in the reference)
 *                    bf16 rows (VQHIP_BF16): the reference evaluates both modes on bf16 TENSORS -- every op of them rounds to bf16 (mode 1:
 *                    x + bf16(q - x); mode 2: norms, quotients, the two dot products, their outer products, the subtraction, the
 *                    addition and the final scale, vqp.py:287-318) -- and so do these kernels, op by op: the value equals the
 *                    reference's bit for bit (but for the order of a row's fp32 sum inside a reduction), which the residual loop
 *                    depends on (rvq.py:524 subtracts it).  The fp32 formula alone would be ~2 % of |out| away.
 *  vqhip_route_bwd : grad_x = J^T g_out (J of the mode; mode 0 = no g_out term; bf16: the frame u, qh, w, |q|/|x| rounded as above)
 *                             + 2 * (*loss_coef) * (x - q) on rows with row_mask != 0  (loss_coef nullable,
 *                             a DEVICE scalar = d loss / d sum_of_squares).
 *                    masked_rows: what the forward left on the rows with row_mask == 0 -- 0: the routed value like any row (the mask
 *                    only keeps them out of the loss), 1: x itself (vqhip_mask_fill_rows: grad_x = g_out there), 2: zeros (grad_x = 0).
 *  vqhip_mask_fill_rows : the padding of a masked batch, in place (vqp.py:1386-1394: quantize = where(mask, quantize, orig_input or
 *                    zeros), indices = where(mask, indices, -1)): rows with row_mask == 0 of q take x's row (zeros != 0: zeros), their
 *                    idx entry -1; q or idx may be null.  Touches the padding rows only. */
int vqhip_route_fwd(const void *x, const void *q, int dtype, int64_t N, int D, int64_t ldx, int64_t ldq,
                    void *out, int64_t ldo, int mode, void *stream);
int vqhip_route_bwd(const void *x, const void *q, const void *g_out, int dtype, int64_t N, int D,
                    int64_t ldx, int64_t ldq, int64_t ldg, const float *loss_coef, const uint8_t *row_mask,
                    int mode, int masked_rows, void *grad_x, int64_t ldo, void *stream);
int vqhip_mask_fill_rows(void *q, const void *x, int dtype, int64_t N, int D, int64_t ldq, int64_t ldx, const uint8_t *row_mask,
                         int64_t *idx, int64_t idx_stride, int zeros, void *stream);

/* ---- gradient routing through the residual loop -------------------------------------------------
 * Replaces, for ResidualVQ.forward with an input that requires grad (rvq.py:469-568, quant_grad_frac = 0): per stage the
 * straight-through / rotation-trick value and Jacobian (vqp.py:282-318, 1225-1233), `quantized_out += quantized` (rvq.py:525),
 * `residual = residual - quantized.detach()` (rvq.py:524) and the backward of every stage's F.mse_loss (vqp.py:1327).
 *  x, g_out, out: [N, D] rows of one dtype; embed: fp32 [Q][C, D] at stride embed_qstride elements (0 = one shared codebook);
 *  idx: int64 [N, idx_stride], the first Q columns are the stages (a negative entry ends the row's loop: dropped quantizers).
 *  backward == 0 : out = sum_q route(r_q, c_q)          (mode 0: plain sum of the codes, 1: straight-through, 2: rotation trick)
 *  backward != 0 : out = sum_q J_q^T g_out (mode 1, 2; g_out null = zero) + 2 * loss_coef[q] * (r_q - c_q) on rows with row_mask != 0;
 *                  loss_coef (nullable): Q DEVICE floats, d loss / d (sum of squared errors of stage q).
 *  resid_routed  : how r_{q+1} is re-derived from r_q.  != 0 (and mode 1 / 2): r_q - route(r_q, c_q), the reference's
 *                  `residual - quantized.detach()` when the layer returned the ROUTED value (training with an input that
 *                  requires grad, vqp.py:1225-1233) -- this is what the indices were searched with (vqhip_chain_t.route_mode);
 *                  0: r_q - c_q (no-grad forward: `quantized` is the code row). */
int vqhip_rvq_route(const void *x, int dtype, int64_t N, int D, int64_t ldx, const float *embed, int64_t embed_qstride,
                    int C, const int64_t *idx, int64_t idx_stride, int Q, int mode, int resid_routed, const void *g_out, int64_t ldg,
                    const float *loss_coef, const uint8_t *row_mask, int backward, void *out, int64_t ldo, void *stream);

/* vqhip_route_fwd / vqhip_route_bwd with q gathered by index from a code table (codes [C, D] in the rows' dtype, contiguous; q[n] =
 * codes[idx[n * idx_stride]]): the [N, D] q tensor is neither written by the search nor read here.  Arithmetic unchanged. */
int vqhip_route_fwd_gather(const void *x, const void *codes, const int64_t *idx, int64_t idx_stride, int dtype, int64_t N, int D,
                           int64_t ldx, void *out, int64_t ldo, int mode, void *stream);
int vqhip_route_bwd_gather(const void *x, const void *codes, const int64_t *idx, int64_t idx_stride, const void *g_out, int dtype,
                           int64_t N, int D, int64_t ldx, int64_t ldg, const float *loss_coef, const uint8_t *row_mask,
                           int mode, int masked_rows, void *grad_x, int64_t ldo, void *stream);

/* out[n] = x[n] - route(x[n], codes[idx[n * idx_stride]]): the input of the next ResidualVQ stage when the layer returned the ROUTED
 * value (`residual - quantized.detach()`, rvq.py:524, in a training step whose input requires grad, vqp.py:1225-1233).
 * mode 1 / 2 and arithmetic as vqhip_route_fwd (bit for bit; bf16 rows: the routed value rounded to bf16 -- the tensor the layer
 * returned -- then the difference).  x, out, codes [C, D] contiguous: in `dtype` (for bf16 rows the bf16 copy of the code rows). */
int vqhip_route_residual(const void *x, int dtype, int64_t N, int D, int64_t ldx, const void *codes, const int64_t *idx, int64_t idx_stride,
                         int mode, void *out, int64_t ldo, void *stream);

/* sum of `n` doubles times `scale` -> one fp32 (commit loss = scale * sum of partials). */
int vqhip_reduce_partials(const double *partials, int64_t n, double scale, float *out, void *stream);
/* R rows of partials in one launch (the per-stage losses of a residual VQ): out[r] = scale * sum(partials[r * stride .. + n)). */
int vqhip_reduce_partials_rows(const double *partials, int R, int64_t n, int64_t stride, double scale, float *out, void *stream);

/* Statistics and folds of H heads in one set of launches (grid dimension y = head), for multi-head modules with separate codebooks
 * (vqp.py:1044-1049: the reference's einsums carry the head axis).  x [H, N, D] at x_hstride elements between heads, idx [H, N], stats
 * [H, stats_stride] floats = embed_sum [C, D] || count [C] per head -- ACCUMULATED INTO, zero it first; it is the buffer a data-parallel
 * caller all-reduces, once for all heads.  workspace: H x vqhip_ema_batched_ws_stride(N, C) bytes.  packed (vqhip_pack_codebook_batched)
 * / embed [H, C, D] / sqerr_partial [H, vqhip_ema_sqerr_partials(N, C)] may be null together (no loss).  Euclidean, or cosine on
 * unit-norm rows.  vqhip_ema_finalize_batched: vqhip_ema_finalize for the H codebooks as the module stores them (cluster_size [H, C],
 * embed_avg / embed [H, C, D]), denom_ws [H, C]. */
size_t vqhip_ema_batched_ws_stride(int64_t N, int C);
int vqhip_ema_accumulate_batched(const void *x, int x_dtype, int H, int64_t N, int D, int64_t ldx, int64_t x_hstride,
                                 const int64_t *idx, const uint8_t *row_mask, int C, float *stats, int64_t stats_stride,
                                 void *workspace, size_t workspace_bytes, const float *packed, const float *embed,
                                 double *sqerr_partial, void *stream);
int vqhip_ema_finalize_batched(float *cluster_size, float *embed_avg, float *embed, const float *stats, int64_t stats_stride,
                               int H, int C, int D, float one_minus_decay, float eps, int cosine, int do_update_ema,
                               float *denom_ws, void *stream);

/* The statistics of S consecutive stages of a residual VQ in one set of launches (grid dimension y = stage; reference: the S calls of
 * Codebook.forward's update block, vqp.py:599-617, one per quantizer of ResidualVQ.forward's loop, rvq.py:469-568).  Stage s: input rows at
 * x + s * x_sstride elements ([N, D] at row stride ldx: the stage inputs of a residual chain, one behind the other), codes in column s of
 * idx [N, idx_stride], statistics ACCUMULATED INTO stats + s * stats_stride (embed_sum [C, D] || count [C]), workspace slice s of
 * vqhip_ema_batched_ws_stride(N, C) bytes (hist_zeroed != 0: the caller zeroed the first C ints of every slice), loss partials
 * sqerr_partial + s * sqerr_stride (nullable with packed / embed).  packed_sstride / embed_sstride: floats between the stages'
 * codebooks (0: one shared codebook, rvq.py:302-306).  Euclidean. */
int vqhip_ema_accumulate_stages(const void *x, int x_dtype, int S, int64_t N, int D, int64_t ldx, int64_t x_sstride,
                                const int64_t *idx, int64_t idx_stride, const uint8_t *row_mask, int C, float *stats,
                                int64_t stats_stride, void *workspace, size_t workspace_bytes, int hist_zeroed,
                                const float *packed, int64_t packed_sstride, const float *embed, int64_t embed_sstride,
                                double *sqerr_partial, int64_t sqerr_stride, void *stream);

/* ---- channel-first layouts ------------------------------------------------------------------------
 * `channel_last = False` / `accept_image_fmap` callers hand over [b, d, n] and the reference rearranges to [b, n, d] and back
 * (vqp.py:1136-1147, 1375-1384).  in [B, R, S] with the batches in_bstride >= R * S elements apart (a channel group of a wider
 * map: GroupedResidualVQ) -> out [B, S, R] contiguous, elements of 2 or 4 bytes copied as bits; 16-byte accesses on both sides when
 * R, S and in_bstride are multiples of 16 / elem_bytes and the pointers 16-byte aligned.  B <= 65535. */
int vqhip_transpose_batched(const void *in, void *out, int elem_bytes, int64_t B, int64_t R, int64_t S, int64_t in_bstride, void *stream);

/* ---- fused train step ---------------------------------------------------------------------------
 * One call = one training forward of an EMA codebook (VectorQuantize.forward in training mode, vqp.py:1176 ->
 * Codebook.forward :673-800; use_cosine_sim: rows normalised by the caller as at :1157-1159, dot-product scores :740-741):
 * pack the codebook, nearest-code search (screened, bit-identical indices), gather q,
 * EMA statistics (count / embed_sum, vqp.py:602-606), the commitment loss' squared error (vqp.py:1327) and -- with fold != 0 --
 * ema_inplace of cluster_size and embed_avg and update_ema (vqp.py:610-617, 576-584).  The same kernels as vqhip_pack_codebook +
 * vqhip_assign_screened + vqhip_ema_accumulate_sqerr + vqhip_ema_finalize + vqhip_reduce_partials, minus what only exists between
 * separate calls: one zeroing kernel for every counter / accumulator, cluster_size folded inside the statistics' scan kernel,
 * embed_avg / embed / loss in one tail kernel (12 launches, was 19).
 * fold == 0 stops after the statistics (data parallel: all-reduce `stats`, then vqhip_ema_finalize).
 * Requirements: vqhip_vq_step_supported(); x rows 16-byte aligned; no dead-code replacement inside (caller's). */
typedef struct {
    const void *x; int64_t x_dtype; int64_t N; int64_t D; int64_t ldx;
    float *embed; float *embed_avg; float *cluster_size; int64_t C;       /* [C, D], [C, D], [C]; updated in place when fold != 0 */
    int64_t *idx_out;                                                     /* [N] */
    void *q_out; int64_t ldq;                                             /* nullable; x's dtype */
    float *stats;                                                         /* [C * D + C]: embed_sum || count of THIS batch (zeroed here) */
    float *loss_out;                                                      /* nullable: loss_scale * sum (q - x)^2 */
    double loss_scale;
    float *packed;                                                        /* vqhip_packed_bytes(C, D) bytes, 16-byte aligned */
    void *workspace; size_t workspace_bytes;                              /* vqhip_vq_step_workspace_bytes(N, C), 256-byte aligned */
    double one_minus_decay; double eps;                                   /* (1 - decay) as the fp32 value ATen uses, Laplace eps */
    int64_t fold;
    void *ev_search_begin; void *ev_search_end;                           /* nullable hipEvent_t: recorded on `stream` around the search
                                                                             (screen + exact passes) -- bench.py's roofline measurement */
    int64_t metric;                                                       /* VQHIP_EUCLID, or VQHIP_COSINE_PRENORM: x holds unit-norm rows
                                                                             (vqhip_l2norm_rows, vqp.py:1159) of a cosine codebook -- the
                                                                             fold then l2-normalises embed (vqp.py:581-582) */
    const uint8_t *row_mask;                                              /* nullable [N]: rows with 0 (padding: `mask` / `lens`,
                                                                             vqp.py:108-110, 599-601) are searched and gathered like the
                                                                             others but leave the statistics and the loss alone */
    /* Row pipeline (round 5).  chunks > 1 (<= 4) with side_stream given: the rows are split into `chunks` contiguous chunks of
     * vqhip_vq_step_chunk_rows(N, chunks) rows; the statistics of chunk k (histogram, scan, scatter, segmented sum + loss: HBM-bound)
     * run on side_stream beside the search of chunk k + 1 on `stream`; the last chunk's statistics and the fold follow on `stream`
     * once the side stream has been joined.  Counts are integers and the sums fp32 atomics in any order already, so the results are
     * those of one chunk up to that rounding; indices / q are identical.  events[0 .. chunks): hipEvent_t of the caller's (the
     * library creates nothing), re-recorded by every call.  chunks <= 1 or side_stream null: everything on `stream`.
     * Chunk k's screening workspace (header: [0] rows of its exact sweep, [1] rows decided between two codes) starts
     * sum_{j<k} (align256(vqhip_screen_workspace_bytes(n_j)) + align256(vqhip_ema_workspace_bytes(n_j, C))) bytes into `workspace`. */
    int64_t chunks;
    void *side_stream;
    void *events[4];
} vqhip_vq_step_t;
int vqhip_vq_step_supported(int x_dtype, int64_t N, int D, int C);
size_t vqhip_vq_step_workspace_bytes(int64_t N, int C);
int64_t vqhip_vq_step_chunk_rows(int64_t N, int chunks);
int vqhip_vq_train_step(const vqhip_vq_step_t *step, void *stream);

/* ---- EMA sufficient statistics ----------------------------------------------------------------
 * Replaces embed_onehot.sum(1) and einsum('h n d, h n c -> h c d') (vqp.py:602, 605).
 * count [C] and embed_sum [C, D] fp32 are ACCUMULATED INTO (zero them first, e.g. hipMemsetAsync).
 * Rows with idx outside [0, C) or row_mask == 0 are skipped.  For metric == VQHIP_COSINE the rows are
 * divided by rnorm[n] (the value vqhip_assign wrote) before accumulation, i.e. the l2-normalised input.
 * Implementation: counting sort of the row ids by code (LDS integer histograms + one global atomic per
 * workgroup and code), then one wave per (code, <=256-row chunk) streams whole rows and adds its partial
 * sum with global fp32 atomics.  workspace: vqhip_ema_workspace_bytes(N, C) bytes, 256-byte aligned. */
size_t vqhip_ema_workspace_bytes(int64_t N, int C);
int vqhip_ema_accumulate(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                         const int64_t *idx, int64_t idx_stride, const float *rnorm, int metric,
                         const uint8_t *row_mask, int C,
                         float *count, float *embed_sum, void *workspace, size_t workspace_bytes,
                         void *stream);

/* The same pass, which reads every unmasked row next to its code, also summing the commitment loss' squared error
 * (reference: F.mse_loss(quantize.detach(), x), vector_quantize_pytorch.py:1327) -- so that the search does not have to re-read
 * x for it.  sqerr_partial[i], i < vqhip_ema_sqerr_partials(N, C): sum over the rows of work item i of ||q - x||^2 with q the
 * row's code in x's dtype (fp32 rows: embed; bf16 rows: the bf16 copy inside `packed`) -- fp32 FMA chains over a few rows'
 * elements, then double: equal to what vqhip_assign / vqhip_assign_screened sum into their own sqerr_partial for the same rows
 * to ~1e-8 relative.  Euclidean metric, D % 4 == 0, D <= 512, 16-byte aligned rows (VQHIP_EINVAL otherwise). */
int64_t vqhip_ema_sqerr_partials(int64_t N, int C);
int vqhip_ema_accumulate_sqerr(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                               const int64_t *idx, int64_t idx_stride, const uint8_t *row_mask, int C,
                               float *count, float *embed_sum, void *workspace, size_t workspace_bytes,
                               const float *packed, const float *embed, double *sqerr_partial, void *stream);
/* vqhip_ema_accumulate[_sqerr] for a caller that has zeroed the histogram (the first C ints of `workspace`) itself, e.g. for all stages
 * of a residual VQ in one launch; packed / embed / sqerr_partial may be null together.  Euclidean. */
int vqhip_ema_accumulate_prezeroed(const void *x, int x_dtype, int64_t N, int D, int64_t ldx,
                               const int64_t *idx, int64_t idx_stride, const uint8_t *row_mask, int C,
                               float *count, float *embed_sum, void *workspace, size_t workspace_bytes,
                               const float *packed, const float *embed, double *sqerr_partial, void *stream);

/* ---- EMA fold + codebook renormalisation ------------------------------------------------------
 * Replaces ema_inplace x2 (vqp.py:76-97, ATen lerp_ semantics), laplace_smoothing + update_ema
 * (:152-154, :576-584).  In place on cluster_size [C], embed_avg [C, D], embed [C, D].
 *   do_lerp        fold count / embed_sum into cluster_size / embed_avg with weight
 *                  w = (1 - decay) * (weight ? weight[c] : 1).
 *   do_update_ema  embed = embed_avg / ((cs + eps) / (sum cs + C eps) * sum cs) [, l2norm if cosine].
 * denom_ws: caller workspace of C floats. */
int vqhip_ema_finalize(float *cluster_size, float *embed_avg, float *embed,
                       const float *count, const float *embed_sum, const float *weight,
                       int C, int D, float one_minus_decay, float eps, int cosine,
                       int do_lerp, int do_update_ema, float *denom_ws, void *stream);

/* vqhip_ema_finalize_batched for H codebooks whose buffers are separate allocations (round 6): the layers of a residual VQ / the groups of
 * GroupedResidualVQ keep one cluster_size / embed_avg / embed buffer set per layer (state_dict keys layers.{i}._codebook.*,
 * residual_vq.py:124-126, 690-691), and the reference folds them one layer after the other (vqp.py:610-617 inside every layer's forward).
 * table: H x 3 DEVICE pointers (uintptr_t) -- cluster_size [C], embed_avg [C, D], embed [C, D] of head h -- in device memory; head h's
 * statistics at stats + h * stats_stride (embed_sum [C, D] || count [C]).  Three launches for all H folds.  denom_ws [H, C]. */
int vqhip_ema_finalize_table(const void *table, const float *stats, int64_t stats_stride, int H, int C, int D,
                             float one_minus_decay, float eps, int cosine, int do_update_ema, float *denom_ws, void *stream);

/* update_ema (vqp.py:576-584) on ONE SHARD of a codebook partitioned over ranks: the Laplace smoothing (vqp.py:152-154) needs
 * sum(cluster_size) and the code count of the whole codebook -- total_cluster_size is a DEVICE scalar (the caller all-reduces the
 * shards' sums), C_total the global code count.  embed = embed_avg / ((cs + eps) / (total + C_total eps) * total) [, l2norm]. */
int vqhip_ema_renormalize_shard(const float *cluster_size, float *embed_avg, float *embed, int C, int D, float eps,
                                const float *total_cluster_size, int C_total, int cosine, float *denom_ws, void *stream);

/* The Q folds of a codebook shared by the stages of a residual VQ (residual_vq.py:213-217: every stage's update_codebook lerps
 * its statistics into the one codebook, vqp.py:616-617; the renormalisation update_ema runs once afterwards, rvq.py:593-598) in
 * one call: the same lerps in the same (stage) order.  stats: Q blocks of `stride` floats, each embed_sum [C, D] then count [C]. */
int vqhip_ema_fold_many(float *cluster_size, float *embed_avg, float *embed, const float *stats, int Q, int64_t stride,
                        int C, int D, float one_minus_decay, float eps, int cosine, int do_update_ema, float *denom_ws,
                        void *stream);

/* ---- decode -----------------------------------------------------------------------------------
 * Replaces codebook[indices] (vqp.py:1003) and get_at('q [c] d, b n q -> q b n d') + sum over q
 * (rvq.py:341-381).  out[n, :] = sum_q embed_q[idx[n, q], :], idx < 0 contributing zero.
 * embed: Q codebooks [C, D] each at stride embed_qstride floats (0 => shared codebook). */
int vqhip_decode_sum(const int64_t *idx, int64_t N, int Q, const float *embed, int64_t embed_qstride,
                     int C, int D, void *out, int out_dtype, int64_t ldo, void *stream);
/* The same running sum over a RANGE of stages (round 6): idx rows idx_stride elements apart (the first Q columns of a wider index
 * tensor; pass idx + q0 and embed + q0 * embed_qstride to start at stage q0), accumulate != 0: the sum continues from what `out` holds
 * (fp32).  Decoding stages [0, Q - 1) while the last stage is still being searched and adding stage Q - 1 afterwards performs the
 * additions of residual_vq.py:525 in the same order -- bit-identical to one call over all stages.  D <= 512. */
int vqhip_decode_sum_range(const int64_t *idx, int64_t idx_stride, int64_t N, int Q, const float *embed, int64_t embed_qstride,
                           int C, int D, void *out, int out_dtype, int64_t ldo, int accumulate, void *stream);

/* ---- ATen-order row sum of squares (vqp.py:59) -- exposed for tests / odd D -------------------- */
/* The K best codes per row, K <= 8 (reference: `logits.topk(topk)` on the N x C `dist` tensor, vector_quantize_pytorch.py:137-138,
 * used by forward(topk=) and ResidualVQ's beam search): scores in the reference's arithmetic (-cdist with correctly rounded sqrt, or
 * cosine similarity), order = (score descending, code ascending); `dist` is never materialised.  idx_out [N, K] int64,
 * val_out nullable [N, K] fp32.  D in {32, 64, 128, 256, 512}; metric VQHIP_EUCLID / VQHIP_COSINE / VQHIP_COSINE_PRENORM. */
int vqhip_topk(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed, int C, int metric, int K,
               int64_t *idx_out, float *val_out, void *stream);

/* Dead-code replacement without a host round trip (reference: Codebook.expire_codes_ / replace,
 * vector_quantize_pytorch.py:544-574, which reads `torch.any(expired)` and `mask.sum().item()` on the host).  The j-th code
 * with cluster_size < threshold (ascending code order) takes candidates[j]: embed[c] = cand[j], cluster_size[c] = reset,
 * embed_avg[c] = cand[j] * reset.  candidates [C, D] fp32: rows the caller drew from torch's generator (randperm(n)[:C],
 * l2-normalised for the cosine metric).  n_expired_out (nullable, device int) receives the number of replaced codes. */
int vqhip_expire_scatter(float *cluster_size, float *embed_avg, float *embed, const float *candidates, int C, int D,
                         float threshold, float reset, int *n_expired_out, void *stream);
/* Dead-code replacement in ONE launch, no candidate tensor (vqp.py:544-574): every code with cluster_size < threshold takes row pi(c) of
 * the batch rows [n, D] (fp32 / bf16, row stride ldx), pi = the affine permutation (a c + b) mod p of Z_p cycle-walked into [0, n),
 * followed by a 4-round Feistel permutation of [0, n) keyed by (a, b) (round 5: the affine map alone hands neighbouring codes rows a
 * fixed stride apart) -- distinct rows for distinct codes when C <= n (the reference samples without replacement, :180-188) -- l2-normalised for a cosine
 * codebook (:545-546); embed_avg = row * reset, cluster_size = reset.  ab: device int64[2] = (a, b), both in [1, p); p: a prime with
 * n <= p < 2^31 (the caller's: the smallest one).  Nothing on the host depends on how many codes expired. */
int vqhip_expire_pick(float *cluster_size, float *embed_avg, float *embed, const void *rows, int x_dtype, int64_t n, int64_t ldx,
                      const int64_t *ab, int64_t p, int C, int D, float threshold, float reset, int cosine, void *stream);

/* k-means centroid update of one iteration (reference: kmeans, vector_quantize_pytorch.py:262-276): in place,
 * means[c] = embed_sum[c] / count[c] where count[c] > 0 (l2-normalised if cosine), unchanged for empty bins. */
int vqhip_kmeans_update(float *means, const float *embed_sum, const float *count, int C, int D, int cosine, void *stream);

/* Score of ONE given code per row in the reference's arithmetic: cdist(x_n, embed[idx_n]) (vector_quantize_pytorch.py:58-62)
 * for VQHIP_EUCLID, the similarity x_n . embed[idx_n] (:741) for VQHIP_COSINE_PRENORM (rows already unit-norm) -- bit for bit
 * the winner's score vqhip_assign reports through best_out.  Replaces nothing in the reference (it always has the whole
 * `dist` tensor); it exists for the codebook-sharded argmin (vector_quantize_pytorch_amd/parallel.py), which merges the
 * shards' winners by (score, index) after a screened search that does not produce scores.  D % 4 == 0, D <= 512. */
int vqhip_score_indices(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, const float *packed, const float *embed,
                        int C, int metric, const int64_t *idx, float *out, void *stream);

/* K11 helper of the codebook-sharded argmin (vq_pack_best_u64 / unpack; no counterpart in the reference, which only
 * replicates codebooks -- its tie rule is ATen argmax's first occurrence, vector_quantize_pytorch.py:140).
 * vqhip_pack_best: key_out[n] = (sortable(negate ? -best[n] : best[n]) << 32) | (0xFFFFFFFF - (idx[n] + index_offset)) as int64:
 *   key order == (score to MAXIMISE, then LOWER global index), so ONE all_reduce(MAX) of N x 8 bytes over the shards yields the
 *   global winner.  best: the shard's winning distance (negate = 1, Euclidean) or similarity (negate = 0); idx: index inside the
 *   shard; index_offset: the shard's first global code.  Global indices must stay below 2^32.
 * vqhip_unpack_best: after the reduction -- gidx_out[n] (nullable) the winning global index, local_out[n] (nullable) that index
 *   relative to own_lo if this rank owns it (own_lo <= g < own_hi) else -1 (what vqhip_decode_sum / vqhip_ema_accumulate skip),
 *   best_out[n] (nullable) the winning score with the sign of the input restored.  One launch each. */
int vqhip_pack_best(const float *best, const int64_t *idx, int64_t N, int64_t index_offset, int negate, int64_t *key_out,
                    void *stream);
int vqhip_unpack_best(const int64_t *key, int64_t N, int64_t own_lo, int64_t own_hi, int negate, int64_t *gidx_out,
                      int64_t *local_out, float *best_out, void *stream);

/* A separate codebook per row (QINCo's implicit neural codebook; reference: Codebook.forward(codebook_transform_fn=),
 * vector_quantize_pytorch.py:729-738: dist = -F.pairwise_distance(x[..., None, :], transformed) or the cosine einsum, then argmax).
 * x [N, D] fp32 at row stride ldx, codes [N, C, D] fp32 contiguous: idx_out[n] = argmin_c ||x_n - codes[n, c] + 1e-6||_2
 * (VQHIP_EUCLID) or argmax_c x_n . codes[n, c] (VQHIP_COSINE_PRENORM: both sides already unit-norm); first extremum in
 * ascending c.  Streams N * C * D floats once. */
int vqhip_assign_rowwise(const float *x, int64_t N, int D, int64_t ldx, const float *codes, int C, int metric,
                         int64_t *idx_out, void *stream);

int vqhip_row_sumsq(const void *x, int x_dtype, int64_t N, int D, int64_t ldx, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VQHIP_H */
