// vq_rvq_chain.hip -- the residual loop of ResidualVQ.forward (rvq.py:469-568) as ONE library call: every launch of the Q chained
// screened searches (vq_screen.hip), of the routed residuals of a gradient step (vqhip_route_residual) and of the per-stage EMA
// statistics (vqhip_ema_accumulate_prezeroed), issued from C on the caller's streams.  Host code only: the kernels are the ones the
// per-stage entry points launch; what this call removes is the host work between them (a Python loop enqueued a cfg-3 forward in
// 0.97 ms -- 1.53 ms with three row chunks -- of a 2.8 ms step, a cfg-5 forward in 5.9 ms of 14.3: tools/host_overhead.py).
//
// Structure (the one the Python loop had):
//   * rows split into K contiguous chunks of whole screening workgroups, chunk k's chain of stages on its own stream (chunk 0: `stream`),
//     so that one chunk's exact passes run beside another chunk's screening kernel;
//   * stage q of a chunk: [routed residual of stage q - 1 -> inputs[q - 1]] , screened search that forms / reads its input and writes
//     its column of idx;
//   * when stage q is final in every chunk (one event per chunk stream), its statistics pass is queued on stats_stream beside the
//     later stages' searches;
//   * on return `stream` has been joined with the chunk streams; the statistics stream is left to the caller to join.
//
// Round 6: G independent loops -- the groups of GroupedResidualVQ (rvq.py:634-724, the loop over `self.rvqs` at :706) -- as ONE launch
// set: every launch above carries blockIdx.y = group (the batched-head forms of the screening kernel, the exact passes and the
// statistics), so a grouped module is Q x (1 screen + 3 exact-pass + 4 statistics launches) instead of G times that on G streams.
#include "vqhip_internal.h"

static inline size_t rc_align(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the 16-byte list header of every (stage, chunk, group) workspace slice and the arrival counters behind it (vq_tail_kernel), in one launch:
// blockIdx.y = slice, the first 4 + n_done ints of each are zeroed
__global__ void __launch_bounds__(256) vq_chain_headers_kernel(char *base, size_t stride, int n_done)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 4 + n_done) ((unsigned *)(base + (size_t)blockIdx.y * stride))[i] = 0u;
}

extern "C" int64_t vqhip_rvq_chain_chunk_rows(int64_t N, int chunks)
{
    if (N <= 0) return 0;
    const int64_t K = chunks < 1 ? 1 : chunks;
    return ((N + K - 1) / K + 255) / 256 * 256;
}

// one screening workspace per (stage, chunk): Q x K slices of this many bytes
extern "C" size_t vqhip_rvq_chain_ws_stride(int64_t N, int chunks)
{
    const int64_t rpc = vqhip_rvq_chain_chunk_rows(N, chunks);
    return rc_align(vqhip_screen_workspace_bytes(rpc < N ? rpc : N), 256);
}

extern "C" int vqhip_rvq_chain_forward(const vqhip_rvq_chain_t *c, void *stream)
{
    if (!c) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: null argument block");
    const int64_t N = c->N, Q = c->Q;
    const int D = (int)c->D, C = (int)c->C, dt = (int)c->x_dtype;
    const int G = c->groups > 1 ? (int)c->groups : 1;
    if (N < 0 || Q < 1 || C < 1) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: bad size");
    if (N == 0) return 0;
    if (!c->x || !c->packed || !c->embed || !c->idx_out || !c->workspace) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: null pointer");
    if (Q > 1 && !c->inputs) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: Q > 1 needs the stage-input buffer");
    if (dt != VQHIP_F32 && dt != VQHIP_BF16) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: unknown dtype");
    const int routed = c->route_mode != 0;
    if (routed && !c->codes) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: routed residuals need the code rows in the rows' dtype");
    if (!routed && Q > 1 && !vqhip_screen_chain_supported(dt, D))
        VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: chained stages take fp32 rows, D in {32, 64, 128, 256} (routed loops: any rows of the screen)");
    if (!vqhip_screen_supported(N, D, C)) VQ_FAIL(VQHIP_EDIM, "rvq_chain_forward: N=%lld D=%d C=%d outside the screened path", (long long)N, D, C);
    int K = (int)c->chunks;
    if (K < 1) K = 1;
    const int64_t rpc = vqhip_rvq_chain_chunk_rows(N, K);
    K = (int)((N + rpc - 1) / rpc);
    if (K > 1 && !c->chunk_streams) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: %d chunks need chunk_streams[0..%d]", K, K - 2);
    const int want_stats = c->stats != nullptr;
    const int stats_side = want_stats && c->stats_stream && c->stats_stream != stream;
    // decode split around the last stage (round 6): sum of the stages [0, Q - 1) on the statistics stream while stage Q - 1 is searched,
    // stage Q - 1 added behind the loop -- the additions of rvq.py:525 in their order; needs the statistics stream and fp32 outputs
    const int split_decode = c->decode_out != nullptr && stats_side && Q >= 2 && dt == VQHIP_F32 && !c->row_mask;
    if (c->decode_out && !split_decode) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: decode_out needs Q >= 2, fp32 rows, no row mask and a statistics stream");
    const int64_t need_ev = (K > 1 || stats_side) ? Q * K + 1 + (split_decode ? 1 : 0) : 0;
    if (need_ev > 0 && (!c->events || c->n_events < need_ev))
        VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: needs %lld events (Q x chunks + 1, + 1 with decode_out)", (long long)need_ev);
    if (split_decode && (c->decode_ldo < D || (G > 1 && c->decode_gstride < D)))
        VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: decode_out needs decode_ldo >= D (and decode_gstride >= D for groups)");
    const size_t wss = vqhip_rvq_chain_ws_stride(N, K);
    if (c->workspace_bytes < wss * (size_t)(Q * K * G)) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: workspace too small");
    if (((uintptr_t)c->workspace) & 255) VQ_FAIL(VQHIP_EALIGN, "rvq_chain_forward: workspace must be 256-byte aligned");
    if (want_stats && (!c->stats_ws || c->stats_ws_stride < vqhip_ema_batched_ws_stride(N, C) || c->stats_stride < (int64_t)C * D + C))
        VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: statistics need stats_ws slices of vqhip_ema_batched_ws_stride(N, C) bytes and stats_stride >= C D + C");
    const int es = dt == VQHIP_BF16 ? 2 : 4;
    if (G > 1) {
        if (c->x_gstride < D || ((c->x_gstride * es) & 15)) VQ_FAIL(VQHIP_EALIGN, "rvq_chain_forward: the groups' rows must be D or more elements apart and stay 16-byte aligned");
        if (c->packed_gstride <= 0 || c->embed_gstride <= 0) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: groups need packed_gstride and embed_gstride");
        if (want_stats && (c->stats_gstride < (int64_t)C * D + C || c->stats_ws_gstride < vqhip_ema_batched_ws_stride(N, C)))
            VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: groups need stats_gstride >= C D + C and stats_ws_gstride >= vqhip_ema_batched_ws_stride(N, C)");
        if (routed && c->codes_gstride <= 0) VQ_FAIL(VQHIP_EINVAL, "rvq_chain_forward: groups with routed residuals need codes_gstride");
    }
    hipStream_t main = (hipStream_t)stream;
    hipError_t e;
    // the Q x K x G list headers (16 bytes each) in one launch
    const int n_done = (int)vq_screen_done_ints(rpc < N ? rpc : N);
    hipLaunchKernelGGL(vq_chain_headers_kernel, dim3((unsigned)((4 + n_done + 255) / 256), (unsigned)(Q * K * G)), dim3(256), 0, main, (char *)c->workspace, wss, n_done);
    if (int rc = vq_launch_status("vq_chain_headers_kernel")) return rc;
    hipEvent_t *ev = (hipEvent_t *)c->events;
    if (K > 1) {                                                   // fork: the chunk streams start behind everything queued on `stream`
        hipEvent_t fork = ev[Q * K];
        if ((e = hipEventRecord(fork, main)) != hipSuccess) VQ_FAIL((int)e, "rvq_chain_forward: hipEventRecord: %s", hipGetErrorString(e));
        for (int k = 1; k < K; ++k)
            if ((e = hipStreamWaitEvent((hipStream_t)c->chunk_streams[k - 1], fork, 0)) != hipSuccess)
                VQ_FAIL((int)e, "rvq_chain_forward: hipStreamWaitEvent: %s", hipGetErrorString(e));
    }
    const char *x0 = (const char *)c->x;
    char *inputs = (char *)c->inputs;
    const size_t in_group = (size_t)N * D * es;                    // bytes between the groups' blocks of one stage input
    const size_t in_stage = in_group * (size_t)G;                  // bytes between consecutive stage inputs ([Q - 1, G, N, D])
    const int64_t idx_g = N * Q;                                   // int64 elements between the groups' index blocks ([G, N, Q])
    const size_t bf16_off = vq_packed_bf16_offset(C, D);

    // stage q's statistics (+ loss partials) for all groups on `ss`
    auto stage_stats = [&](int64_t q, hipStream_t ss) -> int {
        const float *packed_q = c->packed + q * c->packed_qstride;
        const float *embed_q = c->embed + q * c->embed_qstride;
        const void *xin = q == 0 ? c->x : (const void *)(inputs + (size_t)(q - 1) * in_stage);
        float *st_q = c->stats + q * c->stats_stride;
        double *sq_q = c->sqerr_partial ? c->sqerr_partial + q * c->sqerr_stride : nullptr;
        void *ws_q = (char *)c->stats_ws + (size_t)q * c->stats_ws_stride;
        if (G == 1)
            return vqhip_ema_accumulate_prezeroed(xin, dt, N, D, q == 0 ? c->ldx : D, c->idx_out + q, Q, c->row_mask, C, st_q + (size_t)C * D, st_q,
                                                  ws_q, c->stats_ws_stride, sq_q ? packed_q : nullptr, sq_q ? embed_q : nullptr, sq_q, ss);
        const void *qsrc = !sq_q ? nullptr : (dt == VQHIP_BF16 ? (const void *)((const char *)packed_q + bf16_off) : (const void *)embed_q);
        return vq_ema_accumulate_heads(xin, dt, G, N, D, q == 0 ? c->ldx : D, q == 0 ? c->x_gstride * es : (int64_t)in_group, c->idx_out + q, Q,
                                       idx_g * 8, c->row_mask, C, st_q + (size_t)C * D, st_q, c->stats_gstride * 4, ws_q, (int64_t)c->stats_ws_gstride, 1,
                                       qsrc, dt == VQHIP_BF16 ? c->packed_gstride * 4 : c->embed_gstride * 4, sq_q, c->sqerr_gstride * 8, ss);
    };

    for (int64_t q = 0; q < Q; ++q) {
        const float *packed_q = c->packed + q * c->packed_qstride;
        const float *embed_q = c->embed + q * c->embed_qstride;
        for (int k = 0; k < K; ++k) {
            const int64_t r0 = (int64_t)k * rpc, n_k = N - r0 < rpc ? N - r0 : rpc;
            hipStream_t st = k == 0 ? main : (hipStream_t)c->chunk_streams[k - 1];
            vqhip_chain_t ch;
            ch.idx_stride = Q; ch.prev_idx = nullptr; ch.prev_idx_stride = Q; ch.prev_embed = nullptr; ch.x_out = nullptr; ch.ldxo = D;
            ch.route_mode = 0; ch.header_zeroed = 1;
            const void *src = x0 + (size_t)r0 * c->ldx * es;
            int64_t lds = c->ldx;
            int64_t src_g = c->x_gstride * es;                     // bytes between the groups' rows of this stage's search input
            if (q > 0 && routed) {
                // the previous layer returned its ROUTED value and rvq.py:524 subtracted that: its own HBM-bound kernel writes this
                // stage's input, which the search then reads like a first stage's
                const void *psrc = q == 1 ? src : (const void *)(inputs + (size_t)(q - 2) * in_stage + (size_t)r0 * D * es);
                const int64_t plds = q == 1 ? c->ldx : D;
                const int64_t psrc_g = q == 1 ? c->x_gstride * es : (int64_t)in_group;
                void *dst = inputs + (size_t)(q - 1) * in_stage + (size_t)r0 * D * es;
                const void *codes = (const char *)c->codes + (size_t)(q - 1) * c->codes_qstride * es;
                for (int g = 0; g < G; ++g)
                    if (int rc = vqhip_route_residual((const char *)psrc + (size_t)g * psrc_g, dt, n_k, D, plds, (const char *)codes + (size_t)g * c->codes_gstride * es,
                                                      c->idx_out + (int64_t)g * idx_g + r0 * Q + (q - 1), Q, (int)c->route_mode, (char *)dst + (size_t)g * in_group, D, st)) return rc;
                src = dst; lds = D; src_g = (int64_t)in_group;
            } else if (q > 0) {
                ch.prev_idx = c->idx_out + r0 * Q + (q - 1);
                ch.prev_embed = c->embed + (q - 1) * c->embed_qstride;
                ch.x_out = inputs + (size_t)(q - 1) * in_stage + (size_t)r0 * D * es;
                if (q > 1) { src = inputs + (size_t)(q - 2) * in_stage + (size_t)r0 * D * es; lds = D; src_g = (int64_t)in_group; }
            }
            void *ws = (char *)c->workspace + (size_t)(q * K + k) * G * wss;
            if (G == 1) {
                if (int rc = vqhip_assign_screened_chain(src, dt, n_k, D, lds, packed_q, embed_q, C, VQHIP_EUCLID, c->idx_out + r0 * Q + q,
                                                         c->row_mask ? c->row_mask + r0 : nullptr, ws, wss, &ch, st)) return rc;
            } else {
                VqHeadStrides hs;
                hs.heads = G;
                hs.x = src_g; hs.packed = c->packed_gstride * 4; hs.embed = c->embed_gstride * 4;
                hs.codes = dt == VQHIP_BF16 ? hs.packed : hs.embed;
                hs.idx = idx_g * 8; hs.q = 0; hs.ws = (int64_t)wss; hs.xo = (int64_t)in_group;
                if (int rc = vq_assign_screened_impl(src, dt, n_k, D, lds, packed_q, embed_q, C, VQHIP_EUCLID, c->idx_out + r0 * Q + q, nullptr, D, nullptr, D,
                                                     nullptr, c->row_mask ? c->row_mask + r0 : nullptr, ws, wss, nullptr, &ch, 1, st, &hs)) return rc;
            }
            if (need_ev > 0 && (stats_side || q + 1 == Q))
                if ((e = hipEventRecord(ev[q * K + k], st)) != hipSuccess) VQ_FAIL((int)e, "rvq_chain_forward: hipEventRecord: %s", hipGetErrorString(e));
        }
        if (want_stats && stats_side) {
            // stage q's input and indices are final in every chunk: its statistics pass beside the searches of the later stages
            hipStream_t ss = (hipStream_t)c->stats_stream;
            for (int k = 0; k < K; ++k)
                if ((e = hipStreamWaitEvent(ss, ev[q * K + k], 0)) != hipSuccess) VQ_FAIL((int)e, "rvq_chain_forward: hipStreamWaitEvent: %s", hipGetErrorString(e));
            if (int rc = stage_stats(q, ss)) return rc;
            if (split_decode && q == Q - 2) {
                // every stage but the last is final: their part of the output sum, beside the last stage's search
                for (int g = 0; g < G; ++g)
                    if (int rc = vqhip_decode_sum_range(c->idx_out + (int64_t)g * idx_g, Q, N, (int)(Q - 1), c->embed + (int64_t)g * c->embed_gstride,
                                                        c->embed_qstride, C, D, (float *)c->decode_out + (int64_t)g * c->decode_gstride, VQHIP_F32,
                                                        c->decode_ldo, 0, ss)) return rc;
                if ((e = hipEventRecord(ev[Q * K + 1], ss)) != hipSuccess) VQ_FAIL((int)e, "rvq_chain_forward: hipEventRecord: %s", hipGetErrorString(e));
            }
        }
    }
    for (int k = 1; k < K; ++k)                                    // join: `stream` continues behind every chunk's last stage
        if ((e = hipStreamWaitEvent(main, ev[(Q - 1) * K + k], 0)) != hipSuccess) VQ_FAIL((int)e, "rvq_chain_forward: hipStreamWaitEvent: %s", hipGetErrorString(e));
    if (want_stats && !stats_side)
        for (int64_t q = 0; q < Q; ++q)                            // no statistics stream (e.g. under graph capture): behind the loop
            if (int rc = stage_stats(q, main)) return rc;
    if (split_decode) {                                            // + the last stage's codes: the output is complete on `stream`
        if ((e = hipStreamWaitEvent(main, ev[Q * K + 1], 0)) != hipSuccess) VQ_FAIL((int)e, "rvq_chain_forward: hipStreamWaitEvent: %s", hipGetErrorString(e));
        for (int g = 0; g < G; ++g)
            if (int rc = vqhip_decode_sum_range(c->idx_out + (int64_t)g * idx_g + (Q - 1), Q, N, 1,
                                                c->embed + (int64_t)g * c->embed_gstride + (Q - 1) * c->embed_qstride, c->embed_qstride, C, D,
                                                (float *)c->decode_out + (int64_t)g * c->decode_gstride, VQHIP_F32, c->decode_ldo, 1, main)) return rc;
    }
    return 0;
}
