"""ctypes binding of csrc/libvqhip.so (C ABI: include/vqhip.h).

PyTorch is plumbing here: it owns device memory and the stream; every kernel on the hot path is in
the HIP library.  There is NO fallback: if the library is missing or the tensor is not on a GPU the
call raises.
"""
from __future__ import annotations

import ctypes
import functools
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("VQHIP_SO", os.path.join(_HERE, "csrc", "libvqhip.so"))   # VQHIP_SO: A/B experiments only

F32, BF16 = 0, 1
EUCLID, COSINE, COSINE_PRENORM = 0, 1, 2
ASSIGN_ROWS_PER_BLOCK = 128

_lib = None


class VQHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise VQHipError(
            f"{SO_PATH} not found: build it with `make -C {os.path.dirname(SO_PATH)}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "vector_quantize_pytorch_amd has no non-HIP fallback.")
    L = ctypes.CDLL(SO_PATH)
    vp, i64, i32, f32, f64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_double
    L.vqhip_version.restype = ctypes.c_char_p
    L.vqhip_last_error.restype = ctypes.c_char_p
    L.vqhip_packed_bytes.restype = ctypes.c_size_t
    L.vqhip_packed_bytes.argtypes = [i32, i32]
    L.vqhip_assign_blocks.restype = i64
    L.vqhip_assign_blocks.argtypes = [i64]
    L.vqhip_pack_codebook.argtypes = [vp, i32, i32, vp, vp]
    L.vqhip_assign.argtypes = [vp, i32, i64, i32, i64, vp, vp, i32, i32, vp, vp, i32, i64, vp, vp, vp, vp, vp]
    L.vqhip_reduce_partials.argtypes = [vp, i64, f64, vp, vp]
    L.vqhip_reduce_partials_rows.argtypes = [vp, i32, i64, i64, f64, vp, vp]
    L.vqhip_reduce_partials_rows.restype = i32
    L.vqhip_ema_fold_many.argtypes = [vp, vp, vp, vp, i32, i64, i32, i32, f32, f32, i32, i32, vp, vp]
    L.vqhip_ema_fold_many.restype = i32
    L.vqhip_rvq_forward.argtypes = [vp, i32, i64, i32, i64, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp]
    L.vqhip_rvq_forward.restype = i32
    L.vqhip_scores.argtypes = [vp, i32, i64, i32, i64, vp, vp, i32, i32, vp, i64, vp, vp, vp]
    L.vqhip_scores.restype = i32
    L.vqhip_scores_lse.argtypes = [vp, i32, i64, i32, i64, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.vqhip_scores_lse.restype = i32
    L.vqhip_screen_supported.argtypes = [i64, i32, i32]
    L.vqhip_screen_supported.restype = i32
    L.vqhip_screen_workspace_bytes.argtypes = [i64]
    L.vqhip_screen_workspace_bytes.restype = ctypes.c_size_t
    L.vqhip_screen_blocks.argtypes = [i64, i32]
    L.vqhip_screen_blocks.restype = i64
    L.vqhip_screen_partials.argtypes = [i64, i32]
    L.vqhip_screen_partials.restype = i64
    L.vqhip_assign_screened.argtypes = [vp, i32, i64, i32, i64, vp, vp, i32, i32, vp, vp, i64, vp, i64, vp, vp, vp, ctypes.c_size_t, vp, vp]
    L.vqhip_l2norm_rows.argtypes = [vp, i32, i64, i32, i64, vp, i64, vp]
    L.vqhip_l2norm_rows_bwd.argtypes = [vp, vp, i32, i64, i32, i64, i64, vp, i64, vp]
    L.vqhip_transpose_batched.argtypes = [vp, vp, i32, i64, i64, i64, i64, vp]
    L.vqhip_transpose_batched.restype = i32
    L.vqhip_expire_pick.argtypes = [vp, vp, vp, vp, i32, i64, i64, vp, i64, i32, i32, f32, f32, i32, vp]
    L.vqhip_expire_pick.restype = i32
    L.vqhip_mask_fill_rows.argtypes = [vp, vp, i32, i64, i32, i64, i64, vp, vp, i64, i32, vp]
    L.vqhip_mask_fill_rows.restype = i32
    L.vqhip_l2norm_rows_bwd.restype = i32
    L.vqhip_screen_chain_supported.argtypes = [i32, i32]
    L.vqhip_screen_chain_supported.restype = i32
    L.vqhip_assign_screened_chain.argtypes = [vp, i32, i64, i32, i64, vp, vp, i32, i32, vp, vp, vp, ctypes.c_size_t, vp, vp]
    L.vqhip_assign_screened_chain.restype = i32
    L.vqhip_l2norm_rows.restype = i32
    L.vqhip_assign_screened.restype = i32
    L.vqhip_route_fwd.argtypes = [vp, vp, i32, i64, i32, i64, i64, vp, i64, i32, vp]
    L.vqhip_route_bwd.argtypes = [vp, vp, vp, i32, i64, i32, i64, i64, i64, vp, vp, i32, i32, vp, i64, vp]
    L.vqhip_route_fwd.restype = i32
    L.vqhip_ema_renormalize_shard.argtypes = [vp, vp, vp, i32, i32, f32, vp, i32, i32, vp, vp]
    L.vqhip_ema_renormalize_shard.restype = i32
    # (x, dtype, N, D, ldx, embed, embed_qstride, C, idx, idx_stride, Q, mode, g_out, ldg, loss_coef, row_mask, backward, out, ldo, stream)
    L.vqhip_rvq_route.argtypes = [vp, i32, i64, i32, i64, vp, i64, i32, vp, i64, i32, i32, i32, vp, i64, vp, vp, i32, vp, i64, vp]
    L.vqhip_route_bwd.restype = i32
    L.vqhip_ema_workspace_bytes.restype = ctypes.c_size_t
    L.vqhip_ema_workspace_bytes.argtypes = [i64, i32]
    L.vqhip_ema_accumulate.argtypes = [vp, i32, i64, i32, i64, vp, i64, vp, i32, vp, i32, vp, vp, vp, ctypes.c_size_t, vp]
    L.vqhip_ema_sqerr_partials.argtypes = [i64, i32]
    L.vqhip_ema_sqerr_partials.restype = i64
    L.vqhip_ema_accumulate_sqerr.argtypes = [vp, i32, i64, i32, i64, vp, i64, vp, i32, vp, vp, vp, ctypes.c_size_t, vp, vp, vp, vp]
    L.vqhip_ema_accumulate_sqerr.restype = i32
    L.vqhip_ema_accumulate_prezeroed.argtypes = [vp, i32, i64, i32, i64, vp, i64, vp, i32, vp, vp, vp, ctypes.c_size_t, vp, vp, vp, vp]
    L.vqhip_ema_accumulate_prezeroed.restype = i32
    L.vqhip_ema_finalize.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32, i32, i32, vp, vp]
    L.vqhip_decode_sum.argtypes = [vp, i64, i32, vp, i64, i32, i32, vp, i32, i64, vp]
    if hasattr(L, "vqhip_decode_sum_range"):          # (an A/B library of an earlier commit loaded through VQHIP_SO lacks it)
        L.vqhip_decode_sum_range.argtypes = [vp, i64, i64, i32, vp, i64, i32, i32, vp, i32, i64, i32, vp]
        L.vqhip_decode_sum_range.restype = i32
    L.vqhip_row_sumsq.argtypes = [vp, i32, i64, i32, i64, vp, vp]
    L.vqhip_assign_rowwise.argtypes = [vp, i64, i32, i64, vp, i32, i32, vp, vp]
    L.vqhip_assign_rowwise.restype = i32
    L.vqhip_score_indices.argtypes = [vp, i32, i64, i32, i64, vp, vp, i32, i32, vp, vp, vp]
    L.vqhip_score_indices.restype = i32
    L.vqhip_topk.argtypes = [vp, i32, i64, i32, i64, vp, i32, i32, i32, vp, vp, vp]
    L.vqhip_topk.restype = i32
    L.vqhip_expire_scatter.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, vp, vp]
    L.vqhip_expire_scatter.restype = i32
    L.vqhip_kmeans_update.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.vqhip_kmeans_update.restype = i32
    L.vqhip_pack_best.argtypes = [vp, vp, i64, i64, i32, vp, vp]
    L.vqhip_pack_best.restype = i32
    L.vqhip_unpack_best.argtypes = [vp, i64, i64, i64, i32, vp, vp, vp, vp]
    L.vqhip_unpack_best.restype = i32
    L.vqhip_pack_codebook_batched.argtypes = [vp, i32, i32, i32, vp, vp]
    L.vqhip_pack_codebook_batched.restype = i32
    L.vqhip_screen_batched_ws_stride.argtypes = [i64]
    L.vqhip_screen_batched_ws_stride.restype = ctypes.c_size_t
    L.vqhip_assign_screened_batched.argtypes = [vp, i32, i32, i64, i32, i64, i64, vp, vp, i32, i32, vp, vp, i64, i64, vp, vp, ctypes.c_size_t, vp]
    L.vqhip_assign_screened_batched.restype = i32
    L.vqhip_assign_batched.argtypes = [vp, i32, i32, i64, i32, i64, i64, vp, vp, i32, i32, vp, vp, i32, i64, i64, vp, vp, vp]
    L.vqhip_assign_batched.restype = i32
    L.vqhip_route_fwd_gather.argtypes = [vp, vp, vp, i64, i32, i64, i32, i64, vp, i64, i32, vp]
    L.vqhip_route_fwd_gather.restype = i32
    L.vqhip_route_bwd_gather.argtypes = [vp, vp, vp, i64, vp, i32, i64, i32, i64, i64, vp, vp, i32, i32, vp, i64, vp]
    L.vqhip_route_bwd_gather.restype = i32
    L.vqhip_ema_batched_ws_stride.argtypes = [i64, i32]
    L.vqhip_ema_batched_ws_stride.restype = ctypes.c_size_t
    L.vqhip_ema_accumulate_batched.argtypes = [vp, i32, i32, i64, i32, i64, i64, vp, vp, i32, vp, i64, vp, ctypes.c_size_t, vp, vp, vp, vp]
    L.vqhip_ema_accumulate_batched.restype = i32
    L.vqhip_ema_accumulate_stages.argtypes = [vp, i32, i32, i64, i32, i64, i64, vp, i64, vp, i32, vp, i64, vp, ctypes.c_size_t, i32, vp, i64, vp, i64,
                                              vp, i64, vp]
    L.vqhip_ema_accumulate_stages.restype = i32
    L.vqhip_ema_finalize_batched.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, f32, f32, i32, i32, vp, vp]
    L.vqhip_ema_finalize_batched.restype = i32
    L.vqhip_ema_finalize_table.argtypes = [vp, vp, i64, i32, i32, i32, f32, f32, i32, i32, vp, vp]
    L.vqhip_ema_finalize_table.restype = i32
    L.vqhip_route_residual.argtypes = [vp, i32, i64, i32, i64, vp, vp, i64, i32, vp, i64, vp]
    L.vqhip_route_residual.restype = i32
    L.vqhip_vq_step_supported.argtypes = [i32, i64, i32, i32]
    L.vqhip_vq_step_supported.restype = i32
    L.vqhip_vq_step_workspace_bytes.argtypes = [i64, i32]
    L.vqhip_vq_step_workspace_bytes.restype = ctypes.c_size_t
    L.vqhip_rvq_chain_chunk_rows.argtypes = [i64, i32]
    L.vqhip_rvq_chain_chunk_rows.restype = i64
    L.vqhip_rvq_chain_ws_stride.argtypes = [i64, i32]
    L.vqhip_rvq_chain_ws_stride.restype = ctypes.c_size_t
    L.vqhip_rvq_chain_forward.argtypes = [vp, vp]
    L.vqhip_rvq_chain_forward.restype = i32
    L.vqhip_vq_step_chunk_rows.argtypes = [i64, i32]
    L.vqhip_vq_step_chunk_rows.restype = i64
    L.vqhip_vq_train_step.argtypes = [vp, vp]
    L.vqhip_vq_train_step.restype = i32
    for name in ("vqhip_pack_codebook", "vqhip_assign", "vqhip_reduce_partials", "vqhip_ema_accumulate",
                 "vqhip_ema_finalize", "vqhip_decode_sum", "vqhip_row_sumsq", "vqhip_assign_rowwise", "vqhip_score_indices", "vqhip_topk", "vqhip_expire_scatter", "vqhip_kmeans_update", "vqhip_route_fwd", "vqhip_route_bwd", "vqhip_rvq_route", "vqhip_ema_renormalize_shard", "vqhip_scores_lse"):
        getattr(L, name).restype = i32
    _lib = L
    return L


EXPORTS = ("vqhip_version", "vqhip_last_error", "vqhip_packed_bytes", "vqhip_pack_codebook",
           "vqhip_assign_blocks", "vqhip_assign", "vqhip_screen_supported", "vqhip_screen_workspace_bytes",
           "vqhip_screen_blocks", "vqhip_screen_partials", "vqhip_assign_screened", "vqhip_screen_chain_supported", "vqhip_assign_screened_chain", "vqhip_l2norm_rows", "vqhip_l2norm_rows_bwd", "vqhip_transpose_batched", "vqhip_expire_pick", "vqhip_mask_fill_rows", "vqhip_scores", "vqhip_rvq_forward", "vqhip_reduce_partials", "vqhip_reduce_partials_rows", "vqhip_ema_fold_many", "vqhip_ema_workspace_bytes", "vqhip_ema_accumulate", "vqhip_ema_sqerr_partials", "vqhip_ema_accumulate_sqerr",
           "vqhip_ema_finalize", "vqhip_decode_sum", "vqhip_row_sumsq", "vqhip_assign_rowwise", "vqhip_score_indices", "vqhip_topk", "vqhip_expire_scatter", "vqhip_kmeans_update", "vqhip_route_fwd", "vqhip_route_bwd", "vqhip_rvq_route", "vqhip_ema_renormalize_shard", "vqhip_scores_lse",
           "vqhip_pack_best", "vqhip_unpack_best", "vqhip_vq_step_supported", "vqhip_vq_step_workspace_bytes", "vqhip_vq_step_chunk_rows", "vqhip_vq_train_step", "vqhip_route_residual",
           "vqhip_pack_codebook_batched", "vqhip_screen_batched_ws_stride", "vqhip_assign_screened_batched", "vqhip_assign_batched",
           "vqhip_route_fwd_gather", "vqhip_route_bwd_gather", "vqhip_ema_accumulate_prezeroed",
           "vqhip_ema_batched_ws_stride", "vqhip_ema_accumulate_batched", "vqhip_ema_finalize_batched",
           "vqhip_ema_accumulate_stages", "vqhip_rvq_chain_chunk_rows", "vqhip_rvq_chain_ws_stride", "vqhip_rvq_chain_forward",
           "vqhip_ema_finalize_table", "vqhip_decode_sum_range")


def _check(rc, what):
    if rc != 0:
        raise VQHipError(f"{what} failed (rc={rc}): {lib().vqhip_last_error().decode()}")


def _stream():
    # always called inside an @_on_device wrapper: the current device is the tensors' device
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(fn):
    """Run `fn` with the CUDA tensors' device current, so that the stream handed to the library and the kernels it
    launches belong to the device that owns the memory (a module on cuda:1 while cuda:0 is current).  Tensors on
    different GPUs are an error; CPU tensors are rejected by _need_gpu inside `fn`."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = None
        for t in (*args, *kwargs.values()):
            if torch.is_tensor(t) and t.is_cuda:
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise VQHipError(f"{fn.__name__}: tensors on different devices ({dev} and {t.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise VQHipError(f"unsupported dtype {t.dtype}: the HIP path takes float32 or bfloat16")


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise VQHipError("vector_quantize_pytorch_amd runs on MI355X only: got a CPU tensor "
                             "(there is no CPU fallback; move the module and input to 'cuda')")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def as_rows(x: torch.Tensor):
    """View `x` [..., D] as N rows of D elements at a uniform element stride, without copying when
    the layout allows it (contiguous, or a last-dim slice of a contiguous tensor as produced by
    `x.chunk(groups, -1)`).  Returns (tensor_keeping_memory_alive, N, D, ld)."""
    D = x.shape[-1]
    N = x.numel() // D if D else 0
    if x.is_contiguous():
        return x, N, D, D
    if x.stride(-1) == 1 or D == 1:
        lead = [(s, st) for s, st in zip(x.shape[:-1], x.stride()[:-1]) if s != 1]
        if not lead:
            return x, N, D, D
        ld = lead[-1][1]
        ok = ld >= D
        for (s0, st0), (s1, st1) in zip(lead[:-1], lead[1:]):
            ok = ok and (st0 == st1 * s1)
        if ok:
            return x, N, D, ld
    xc = x.contiguous()
    return xc, N, D, D


# ------------------------------------------------------------------------------------------------
# thin wrappers (allocation is done here, in torch; the library never allocates)
# ------------------------------------------------------------------------------------------------
@_on_device
def pack_codebook(embed2d: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    _need_gpu(embed2d)
    assert embed2d.dtype == torch.float32 and embed2d.is_contiguous() and embed2d.ndim == 2
    C, D = embed2d.shape
    nbytes = lib().vqhip_packed_bytes(C, D)
    if nbytes == 0:
        raise VQHipError(f"codebook [{C}, {D}] unsupported: the HIP path handles 1 <= dim <= 2048")
    if out is None or out.numel() * 4 != nbytes:
        out = torch.empty(nbytes // 4, dtype=torch.float32, device=embed2d.device)
    _check(lib().vqhip_pack_codebook(_ptr(embed2d), C, D, _ptr(out), _stream()), "vqhip_pack_codebook")
    return out


@_on_device
def pack_codebook_batched(embed3d: torch.Tensor) -> torch.Tensor:
    """embed [H, C, D] -> packed [H, packed floats]: the H codebooks of a multi-head module in two launches"""
    _need_gpu(embed3d)
    assert embed3d.dtype == torch.float32 and embed3d.is_contiguous() and embed3d.ndim == 3
    H, C, D = embed3d.shape
    nbytes = lib().vqhip_packed_bytes(C, D)
    if nbytes == 0:
        raise VQHipError(f"codebook [{C}, {D}] unsupported: the HIP path handles 1 <= dim <= 2048")
    out = torch.empty(H, nbytes // 4, dtype=torch.float32, device=embed3d.device)
    _check(lib().vqhip_pack_codebook_batched(_ptr(embed3d), H, C, D, _ptr(out), _stream()), "vqhip_pack_codebook_batched")
    return out


def assign_batched_supported(xs: torch.Tensor, C: int, *, cosine=False, skip_l2norm=False) -> bool:
    """can the H searches of xs [H, ..., D] run as ONE launch set (vqhip_assign_screened_batched, or vqhip_assign_batched for the
    dims the screen does not take)?  Needs one uniform row stride for every head and a uniform stride between the heads."""
    if not (xs.is_cuda and xs.ndim >= 3 and xs.shape[0] > 1 and xs.dtype in (torch.float32, torch.bfloat16)):
        return False
    if os.environ.get("VQHIP_SCREEN_VERIFY", "0") == "1" or screen_debug or xs.shape[-1] > 512:
        return False
    xk, N, D, ldx = as_rows(xs[0])
    es = xk.element_size()
    return bool(N > 0 and xk.data_ptr() == xs.data_ptr() and (xs.stride(0) * es) % 16 == 0 and xs.data_ptr() % 16 == 0
                and all(as_rows(xs[h])[3] == ldx and as_rows(xs[h])[0].data_ptr() == xs[h].data_ptr() for h in range(1, xs.shape[0])))


@_on_device
def assign_batched(xs: torch.Tensor, packed: torch.Tensor, embed3d: torch.Tensor, *, cosine=False, skip_l2norm=False, want_q=True,
                   want_rnorm=False, row_mask=None):
    """xs [H, ..., D] (head h's rows h * xs.stride(0) elements behind head 0's), packed [H, P] from pack_codebook_batched, embed [H, C, D]
    -> dict(idx [H, ...], q [H, ..., D] | None, rnorm [H, ...] | None): the search of every head in one set of launches -- screened
    (csrc/vq_screen.hip) when the dim allows it and the rows need no l2norm inside the kernel, the exact kernel otherwise."""
    _need_gpu(xs, packed, embed3d, row_mask)
    H = xs.shape[0]
    xk, N, D, ldx = as_rows(xs[0])
    C = embed3d.shape[1]
    assert embed3d.shape == (H, C, D) and embed3d.is_contiguous() and embed3d.dtype == torch.float32 and packed.is_contiguous()
    dev, lead = xs.device, xs.shape[1:-1]
    idx = torch.empty(H, N, dtype=torch.int64, device=dev)
    q = torch.empty(H, N, D, dtype=xs.dtype, device=dev) if want_q else None
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
        assert row_mask.numel() == N
    es = xk.element_size()
    screened = ((not cosine or skip_l2norm) and not want_rnorm and screening_enabled() and bool(lib().vqhip_screen_supported(N, D, C))
                and (ldx * es) % 16 == 0)
    rnorm = None
    if screened:
        wss = lib().vqhip_screen_batched_ws_stride(N)
        ws = torch.empty(H * wss, dtype=torch.uint8, device=dev)
        _check(lib().vqhip_assign_screened_batched(_ptr(xk), _dtype_code(xk), H, N, D, ldx, xs.stride(0), _ptr(packed), _ptr(embed3d), C,
                                                   COSINE_PRENORM if cosine else EUCLID, _ptr(idx), _ptr(q), D, N * D, _ptr(row_mask),
                                                   _ptr(ws), H * wss, _stream()), "vqhip_assign_screened_batched")
    else:
        need_rn = want_rnorm or cosine or (D % 32 != 0)
        rnorm = torch.empty(H, N, dtype=torch.float32, device=dev) if need_rn else None
        _check(lib().vqhip_assign_batched(_ptr(xk), _dtype_code(xk), H, N, D, ldx, xs.stride(0), _ptr(packed), _ptr(embed3d), C,
                                          (COSINE_PRENORM if skip_l2norm else COSINE) if cosine else EUCLID, _ptr(idx), _ptr(q),
                                          _dtype_code(xs), D, N * D, _ptr(rnorm), _ptr(row_mask), _stream()), "vqhip_assign_batched")
    return dict(idx=idx.view(H, *lead), q=None if q is None else q.view(H, *lead, D), rnorm=None if rnorm is None else rnorm.view(H, *lead))


screen_debug = False   # tests: also return the screening kernel's per-row (best, second, threshold, flagged)


def screening_enabled() -> bool:
    """VQHIP_SCREEN=0 forces the exact fp32-MFMA kernel for bf16 inputs too (A/B measurements, profiling)."""
    return os.environ.get("VQHIP_SCREEN", "1") != "0"


@_on_device
def assign(x: torch.Tensor, packed: torch.Tensor, embed2d: torch.Tensor, *, cosine=False,
           want_q=True, want_sqerr=False, want_best=False, want_rnorm=False, row_mask=None, q_out=None,
           skip_l2norm=False, resid_out=None):
    """x [..., D] -> dict(idx [...], q [..., D] | None, sqerr_partials | None, best, rnorm).
    resid_out (optional, x's shape and dtype, rows may be strided) receives x - q, the next residual-VQ stage's input."""
    _need_gpu(x, packed, embed2d, row_mask)
    xk, N, D, ldx = as_rows(x)
    C = embed2d.shape[0]
    assert embed2d.shape[1] == D and embed2d.is_contiguous() and embed2d.dtype == torch.float32
    dev = x.device
    lead = x.shape[:-1]
    idx = torch.empty(lead, dtype=torch.int64, device=dev)
    q = None
    ldq = D
    if want_q:
        q = q_out if q_out is not None else torch.empty(*lead, D, dtype=x.dtype, device=dev)
        assert q.is_contiguous() and q.dtype == x.dtype
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
        assert row_mask.numel() == N
    screened = ((not cosine or skip_l2norm) and not want_best and not want_rnorm and N > 0 and screening_enabled()
                and xk.dtype in (torch.bfloat16, torch.float32) and xk.data_ptr() % 16 == 0
                and (ldx * xk.element_size()) % 16 == 0 and embed2d.data_ptr() % 16 == 0
                and bool(lib().vqhip_screen_supported(N, D, C)))
    if resid_out is not None and not want_q and not screened:
        want_q = True          # the exact kernel has no residual output: x - q is formed below
        q = torch.empty(*lead, D, dtype=x.dtype, device=dev)
    if screened:
        # MFMA screen + exact arithmetic on the uncertified rows only (csrc/vq_screen.hip); same outputs.  n_exact = rows that
        # took the full exact sweep, n_pair = rows decided between two candidate codes (bf16 rows), both device-side counters
        nblk = lib().vqhip_screen_partials(N, _dtype_code(xk))
        partials = torch.empty(nblk, dtype=torch.float64, device=dev) if want_sqerr else None
        nws = lib().vqhip_screen_workspace_bytes(N)
        ws = torch.empty((nws + 3) // 4, dtype=torch.int32, device=dev)
        dbg = torch.empty(N, 4, dtype=torch.float32, device=dev) if screen_debug else None
        rk, ldr = None, 0
        if resid_out is not None:
            rk, rN, rD, ldr = as_rows(resid_out)
            assert rN == N and rD == D and rk.dtype == xk.dtype
        _check(lib().vqhip_assign_screened(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(packed), _ptr(embed2d), C,
                                           COSINE_PRENORM if cosine else EUCLID, _ptr(idx), _ptr(q), ldq,
                                           _ptr(rk), ldr, _ptr(partials), _ptr(row_mask), _ptr(ws), nws, _ptr(dbg), _stream()),
               "vqhip_assign_screened")
        if os.environ.get("VQHIP_SCREEN_VERIFY", "0") == "1":
            # paranoia switch: run the exact fp32-MFMA kernel on the same input as well and insist on identical indices
            # (costs the exact kernel's time plus a host sync; meant for validating the certification on one's own data)
            chk = torch.empty_like(idx)
            rn = torch.empty(lead, dtype=torch.float32, device=dev) if cosine else None
            _check(lib().vqhip_assign(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(packed), _ptr(embed2d), C,
                                      COSINE_PRENORM if cosine else EUCLID, _ptr(chk), None, _dtype_code(x), ldq,
                                      None, _ptr(rn), None, None, _stream()), "vqhip_assign (verify)")
            bad = int((chk != idx).sum().item())
            if bad:
                raise VQHipError(f"VQHIP_SCREEN_VERIFY: screened and exact search disagree on {bad} of {N} rows")
        return dict(idx=idx, q=q, sqerr_partials=partials, best=None, rnorm=None, nblk=nblk, n_exact=ws[:1], n_pair=ws[1:2], screen_debug=dbg)
    need_rn = want_rnorm or cosine or (D % 32 != 0)
    rnorm = torch.empty(lead, dtype=torch.float32, device=dev) if need_rn else None
    best = torch.empty(lead, dtype=torch.float32, device=dev) if want_best else None
    nblk = lib().vqhip_assign_blocks(N)
    partials = torch.empty(max(nblk, 1), dtype=torch.float64, device=dev) if want_sqerr else None
    if N > 0:
        _check(lib().vqhip_assign(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(packed), _ptr(embed2d), C,
                                  (COSINE_PRENORM if skip_l2norm else COSINE) if cosine else EUCLID, _ptr(idx), _ptr(q),
                                  _dtype_code(x), ldq,
                                  _ptr(best), _ptr(rnorm), _ptr(partials), _ptr(row_mask), _stream()),
               "vqhip_assign")
    elif partials is not None:
        partials.zero_()
    if resid_out is not None:
        torch.sub(x, q, out=resid_out)
    return dict(idx=idx, q=q, sqerr_partials=partials, best=best, rnorm=rnorm, nblk=nblk)


@_on_device
def scores(x: torch.Tensor, packed: torch.Tensor, embed2d: torch.Tensor, *, cosine=False, skip_l2norm=False):
    """x [..., D] -> (dist [..., C] fp32 = -cdist or cosine similarity, argmax idx [...], rnorm).  Rare options only."""
    _need_gpu(x, packed, embed2d)
    xk, N, D, ldx = as_rows(x)
    C = embed2d.shape[0]
    out = torch.empty(*x.shape[:-1], C, dtype=torch.float32, device=x.device)
    idx = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device)
    rnorm = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    if N > 0:
        metric = (COSINE_PRENORM if skip_l2norm else COSINE) if cosine else EUCLID
        _check(lib().vqhip_scores(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(packed), _ptr(embed2d), C, metric,
                                  _ptr(out), C, _ptr(idx), _ptr(rnorm), _stream()), "vqhip_scores")
    return out, idx, rnorm


@_on_device
def scores_lse(x: torch.Tensor, packed: torch.Tensor, embed2d: torch.Tensor, target=None, *, cosine=False, skip_l2norm=False):
    """Streaming form of what F.cross_entropy(dist, target) reads from the score row (vqp.py:1242-1256): x [..., D] ->
    (lse [...] = log sum_c exp(dist[.., c]), tscore [...] = dist[.., target] (target None: the winner's score; target < 0: 0),
    argmax idx [...]), dist as in scores().  No N x C tensor."""
    _need_gpu(x, packed, embed2d, target)
    xk, N, D, ldx = as_rows(x)
    C = embed2d.shape[0]
    lead, dev = x.shape[:-1], x.device
    lse = torch.empty(lead, dtype=torch.float32, device=dev)
    ts = torch.empty(lead, dtype=torch.float32, device=dev)
    idx = torch.empty(lead, dtype=torch.int64, device=dev)
    rnorm = torch.empty(lead, dtype=torch.float32, device=dev)
    if target is not None:
        target = target.reshape(-1).to(torch.int64).contiguous()
        assert target.numel() == N
    if N > 0:
        metric = (COSINE_PRENORM if skip_l2norm else COSINE) if cosine else EUCLID
        _check(lib().vqhip_scores_lse(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(packed), _ptr(embed2d), C, metric, _ptr(target),
                                      _ptr(lse), _ptr(ts), _ptr(idx), _ptr(rnorm), _stream()), "vqhip_scores_lse")
    return lse, ts, idx


def screen_supported(x: torch.Tensor, C: int) -> bool:
    """True when assign() would take the screened path for rows like x (Euclidean, or cosine on unit-norm rows)."""
    xk, N, D, ldx = as_rows(x)
    return bool(N > 0 and screening_enabled() and xk.dtype in (torch.bfloat16, torch.float32) and xk.data_ptr() % 16 == 0
                and (ldx * xk.element_size()) % 16 == 0 and lib().vqhip_screen_supported(N, D, C))


@_on_device
def l2norm_rows(x: torch.Tensor) -> torch.Tensor:
    """x / max(||x||, 1e-6) row-wise in the reference's arithmetic (vqp.py:37-38 at :1159); D in {32, 64, 128, 256}."""
    _need_gpu(x)
    xk, N, D, ldx = as_rows(x)
    out = torch.empty(*x.shape, dtype=x.dtype, device=x.device)
    if N > 0:
        _check(lib().vqhip_l2norm_rows(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(out), D, _stream()), "vqhip_l2norm_rows")
    return out


def is_transposed_view(t: torch.Tensor) -> bool:
    """t [B, n, d] whose memory is a [B, d, n] tensor with contiguous [d, n] planes (x.transpose(1, 2), a flattened feature map
    'b d (h w)' transposed, a channel group of one): the layout a channel-first caller's input, and the gradient of its output, arrive in"""
    return (t.ndim == 3 and t.is_cuda and t.element_size() in (2, 4) and t.numel() > 0 and t.shape[0] <= 65535
            and t.stride(1) == 1 and t.stride(2) == t.shape[1] and (t.shape[0] == 1 or t.stride(0) >= t.shape[1] * t.shape[2])
            and t.shape[1] > 1 and t.shape[2] > 1)


def transpose_rows(t: torch.Tensor) -> torch.Tensor:
    """contiguous copy of a transposed view (is_transposed_view): ONE tiled transposing kernel (vqhip_transpose_batched)"""
    _need_gpu(t)
    B, n, d = t.shape
    out = torch.empty(B, n, d, dtype=t.dtype, device=t.device)
    _check(lib().vqhip_transpose_batched(_ptr(t), _ptr(out), t.element_size(), B, d, n, t.stride(0) if B > 1 else d * n, _stream()),
           "vqhip_transpose_batched")
    return out


def rows_contiguous(t: torch.Tensor) -> torch.Tensor:
    """t.contiguous(), through the transposing kernel when t is a transposed view of a channel-first tensor"""
    if t.is_contiguous():
        return t
    if is_transposed_view(t) and os.environ.get("VQHIP_TRANSPOSE", "1") != "0":
        return transpose_rows(t)
    return t.contiguous()


def l2norm_rows_supported(x: torch.Tensor) -> bool:
    """rows l2norm_rows / l2norm_rows_bwd take: float32 / bfloat16 on the GPU, D in {32, 64, 128, 256, 512}, 4-element aligned rows"""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.ndim >= 2 and x.shape[-1] in (32, 64, 128, 256, 512)
            and x.numel() > 0):
        return False
    try:
        xk, N, D, ldx = as_rows(x)
    except Exception:
        return False
    es = xk.element_size()
    return xk.data_ptr() % (4 * es) == 0 and (ldx * es) % (4 * es) == 0


WIDE_MAX_DIM = 2048


def wide_dim(D: int) -> bool:
    """512 < D <= 2048: served by the plain exact kernels of csrc/vq_wide.hip (no screened search, no score-row options, no fused
    residual loop / fused train step; l2norm forward only)"""
    return 512 < int(D) <= WIDE_MAX_DIM


def l2norm_rows_bwd(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """gradient of l2norm_rows(x) with respect to x for the output gradient g (autograd's F.normalize backward, one kernel)"""
    _need_gpu(x, g)
    xk, N, D, ldx = as_rows(x)
    g = g.to(x.dtype)
    if not g.is_contiguous():
        g = g.contiguous()
    out = torch.empty(*x.shape, dtype=x.dtype, device=x.device)
    if N > 0:
        _check(lib().vqhip_l2norm_rows_bwd(_ptr(xk), _ptr(g), _dtype_code(xk), N, D, ldx, D, _ptr(out), D, _stream()), "vqhip_l2norm_rows_bwd")
    return out


class _Chain(ctypes.Structure):          # vqhip_chain_t (include/vqhip.h)
    _fields_ = [("idx_stride", ctypes.c_int64), ("prev_idx", ctypes.c_void_p), ("prev_idx_stride", ctypes.c_int64),
                ("prev_embed", ctypes.c_void_p), ("x_out", ctypes.c_void_p), ("ldxo", ctypes.c_int64), ("route_mode", ctypes.c_int64),
                ("header_zeroed", ctypes.c_int64)]


def rvq_chain_supported(x: torch.Tensor, C: int, routed=False) -> bool:
    """can the residual loop run as a chain (every stage forms its input in its own prologue: vqhip_assign_screened_chain)?
    routed: the loop of a training step whose input requires grad -- every stage's input is written by vqhip_route_residual and
    searched like a first stage, so bf16 rows and D = 512 qualify as well."""
    if not (x.is_cuda and x.dtype in ((torch.float32, torch.bfloat16) if routed else (torch.float32,)) and screening_enabled()
            and os.environ.get("VQHIP_RVQ_CHAIN", "1") != "0"):
        return False
    if os.environ.get("VQHIP_SCREEN_VERIFY", "0") == "1" or screen_debug:
        return False            # the paranoia / debug switches live in assign(): stage-by-stage loop
    xk, N, D, ldx = as_rows(x)
    es = xk.element_size()
    return bool(N > 0 and (routed or lib().vqhip_screen_chain_supported(_dtype_code(xk), D)) and lib().vqhip_screen_supported(N, D, C)
                and xk.data_ptr() % 16 == 0 and (ldx * es) % 16 == 0)


_CACHE_CAP = 32        # entries per (device, caller stream)-keyed cache below: least recently used out first


def _lru(cache: dict, key, make):
    """cache[key], created by make() on a miss; the caches are keyed by the caller stream's raw handle (torch's own streams come from a
    fixed pool and are never destroyed, so a handle is a stable name; every other kind of stream -- ExternalStream -- is why the caches
    are bounded: at most _CACHE_CAP entries, least recently used first out)."""
    v = cache.pop(key, None)
    if v is None:
        v = make()
        while len(cache) >= _CACHE_CAP:
            cache.pop(next(iter(cache)))
    cache[key] = v               # (re-inserted: dicts keep insertion order, the first key is the least recently used)
    return v


_CHAIN_STREAMS = {}


def _chain_streams(device, main, n):
    """n side streams that pair with `main` for the row chunks of a residual chain (per caller stream: concurrent groups do not share)"""
    pool = _lru(_CHAIN_STREAMS, (torch.device(device).index, main.cuda_stream), list)
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def rvq_row_chunks(N: int) -> int:
    """Row chunks the residual chain is split into (each chunk runs its own chain of stages on its own stream, so that one chunk's
    short exact passes -- refine, pair, finish: latency-bound launches every stage has to wait for -- run beside another chunk's
    screening kernel instead of leaving the chip idle).  Measured at cfg 3 (2^18 rows x 8 stages, profiles/r5_rvq_cfg3/chunks.txt):
    1 chunk 2.86-2.89 ms, 2 chunks 2.88, 3 chunks 2.79-2.80, 4 chunks 3.37 (chunks of 2^16 rows = 256 workgroups no longer fill the
    chip's 512 workgroup slots) -- default: chunks of about 87 000 rows, at most 3.  VQHIP_RVQ_CHUNKS overrides; chunks are
    multiples of 256 rows (one screening workgroup) and hold at least 65 536 rows."""
    env = os.environ.get("VQHIP_RVQ_CHUNKS")
    k = int(env) if env else min(3, N // 87381)
    return max(1, min(k, N // 65536))


def grvq_row_chunks(N: int, G: int) -> int:
    """Row chunks of the batched grouped chain (GroupedResidualVQ._forward_batched).  Every launch already carries the G groups'
    workgroups (cfg 5: 4 096 per screening launch, eight rounds of the chip's 512 slots), and splitting the rows on top only adds
    launches: cfg 5 on one box, same process conditions -- 1 chunk 14.04-14.32 ms, 2 chunks 14.22-14.57, 3 chunks 14.60-14.82
    (profiles/r6_ab/summary.md, r6b).  VQHIP_GRVQ_CHUNKS overrides."""
    return 1


@_on_device
def rvq_forward_chained(x: torch.Tensor, packed: torch.Tensor, embed: torch.Tensor, Q: int, *, row_mask=None, stage_hook=None,
                        fill_masked=True, route_mode=0, row_chunks=1):
    """The residual loop (rvq.py:469-568) as Q chained screened searches: stage q's kernel forms its input
    inputs[q-1] - embed[idx[:, q-1]] in its prologue and stores it as inputs[q]; no stage re-reads its input to write a residual,
    and every stage writes its column of idx directly.  No q / squared-error outputs: the caller's statistics pass sums the loss
    (ema_accumulate(sqerr_from=...)).  -> dict(idx [..., Q], inputs [Q tensors], n_exact / n_pair per stage).
    fill_masked=False: the caller writes the -1 of the masked rows itself (mask_fill_indices) -- needed when stage_hook hands `idx`
    to work on ANOTHER stream, which may still be reading it when this function returns.
    route_mode (0 / STRAIGHT_THROUGH / ROTATION): what each layer returned as `quantized`, i.e. what rvq.py:524 subtracted -- the
    code row, or the routed value of a training step whose input requires grad (the arithmetic of route_fwd, bit for bit).
    row_chunks K > 1: the rows are split into K contiguous chunks, each running its own chain on its own stream (chunk 0 on the
    caller's); rows are independent units, so the results are those of K = 1 bit for bit.  stage_hook then receives
    `ready`, one event per chunk stream recorded behind stage q (its consumer waits for all of them); on return the caller's stream
    has been joined with every chunk stream."""
    _need_gpu(x, packed, embed, row_mask)
    shared = embed.ndim == 2
    xk, N, D, ldx = as_rows(x)
    lead, dev = x.shape[:-1], x.device
    C = embed.shape[-2]
    assert embed.dtype == torch.float32 and embed.is_contiguous()
    assert xk.dtype == torch.float32 or (route_mode and xk.dtype == torch.bfloat16), "chained stages: float32 rows"
    dt = _dtype_code(xk)
    es = xk.element_size()
    idx = torch.empty(N, Q, dtype=torch.int64, device=dev)
    bufs = torch.empty(max(Q - 1, 1), N, D, dtype=xk.dtype, device=dev)
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    K = max(1, int(row_chunks))
    rpc = ((N + K - 1) // K + 255) // 256 * 256               # rows per chunk: whole screening workgroups
    starts = [r0 for r0 in range(0, N, rpc)]
    K = len(starts)
    nws = lib().vqhip_screen_workspace_bytes(min(rpc, N))
    nws4 = (nws + 15) // 16 * 4                       # ints per stage and chunk, 16-byte granules
    ws_all = torch.empty(Q, K, nws4, dtype=torch.int32, device=dev)
    n_done = ((min(rpc, N) + 127) // 128 + 1) & ~1    # arrival counters of the merged exact-pass launch (vq_tail_kernel) sit behind the header
    ws_all[:, :, :4 + n_done].zero_()                 # the list headers (+ counters) in ONE launch (a 16-byte memset per stage cost 13 us of gaps each)
    codes = embed if xk.dtype == torch.float32 else embed.to(xk.dtype)      # (bf16 rows: the routing kernel gathers bf16 code rows)
    main = torch.cuda.current_stream(dev)
    streams = [main]
    if K > 1:
        streams += _chain_streams(dev, main, K - 1)
        fork = torch.cuda.Event()
        fork.record(main)
        for s in streams[1:]:
            s.wait_event(fork)
    x_ptr, idx_ptr, mask_ptr = xk.data_ptr(), idx.data_ptr(), (row_mask.data_ptr() if row_mask is not None else None)
    inputs, counts = [x], []
    for q in range(Q):
        ready = []
        for k, r0 in enumerate(starts):
            n_k = min(rpc, N - r0)
            ws = ws_all[q, k]
            with torch.cuda.stream(streams[k]):
                ch = _Chain(idx_stride=Q, prev_idx=None, prev_idx_stride=Q, prev_embed=None, x_out=None, ldxo=D, route_mode=0, header_zeroed=1)
                src, lds = x_ptr + r0 * ldx * es, ldx
                if q > 0 and route_mode:
                    # the previous layer returned its ROUTED value (rotation trick / straight-through on an input that requires grad) and
                    # rvq.py:524 subtracted THAT: an HBM-bound kernel of its own (vqhip_route_residual) writes this stage's input, which
                    # the search then reads like a first stage's
                    prev_c = codes if shared else codes[q - 1]
                    psrc, plds = (src, ldx) if q == 1 else (bufs[q - 2].data_ptr() + r0 * D * es, D)
                    dst = bufs[q - 1].data_ptr() + r0 * D * es
                    _check(lib().vqhip_route_residual(ctypes.c_void_p(psrc), dt, n_k, D, plds, _ptr(prev_c),
                                                      ctypes.c_void_p(idx_ptr + 8 * (r0 * Q + q - 1)), Q, int(route_mode),
                                                      ctypes.c_void_p(dst), D, _stream()), "vqhip_route_residual")
                    src, lds = dst, D
                elif q > 0:
                    prev_e = embed if shared else embed[q - 1]
                    ch.prev_idx = idx_ptr + 8 * (r0 * Q + q - 1)
                    ch.prev_embed = prev_e.data_ptr()
                    ch.x_out = bufs[q - 1].data_ptr() + r0 * D * es
                    if q > 1:
                        src, lds = bufs[q - 2].data_ptr() + r0 * D * es, D
                _check(lib().vqhip_assign_screened_chain(ctypes.c_void_p(src), dt, n_k, D, lds, _ptr(packed if shared else packed[q]),
                                                         _ptr(embed if shared else embed[q]), C, EUCLID,
                                                         ctypes.c_void_p(idx_ptr + 8 * (r0 * Q + q)),
                                                         ctypes.c_void_p(mask_ptr + r0) if mask_ptr is not None else None, _ptr(ws), nws,
                                                         ctypes.byref(ch), _stream()), "vqhip_assign_screened_chain")
                if K > 1 and (stage_hook is not None or q + 1 == Q):
                    ev = torch.cuda.Event()
                    ev.record(streams[k])
                    ready.append(ev)
        if q > 0:
            inputs.append(bufs[q - 1].view(*lead, D))
        counts.append((ws_all[q, :, 0], ws_all[q, :, 1]))      # per chunk: rows of the exact sweep / rows decided between two codes
        if stage_hook is not None:      # stage q's input and indices are final (in stream order): the caller's per-stage work
            if K > 1:
                stage_hook(q, inputs[q], idx.view(*lead, Q), ready=ready)
            else:
                stage_hook(q, inputs[q], idx.view(*lead, Q))
    if K > 1:
        for ev in ready[1:]:
            main.wait_event(ev)
    idx = idx.view(*lead, Q)
    if row_mask is not None and fill_masked:
        mask_fill_indices(idx, row_mask)
    return dict(idx=idx, inputs=inputs, counts=counts, bufs=bufs)


class _RvqChain(ctypes.Structure):       # vqhip_rvq_chain_t (include/vqhip.h)
    _fields_ = [("x", ctypes.c_void_p), ("x_dtype", ctypes.c_int64), ("N", ctypes.c_int64), ("D", ctypes.c_int64), ("ldx", ctypes.c_int64),
                ("packed", ctypes.c_void_p), ("packed_qstride", ctypes.c_int64), ("embed", ctypes.c_void_p), ("embed_qstride", ctypes.c_int64),
                ("C", ctypes.c_int64), ("Q", ctypes.c_int64), ("idx_out", ctypes.c_void_p), ("inputs", ctypes.c_void_p),
                ("row_mask", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("route_mode", ctypes.c_int64), ("codes", ctypes.c_void_p), ("codes_qstride", ctypes.c_int64),
                ("stats", ctypes.c_void_p), ("stats_stride", ctypes.c_int64), ("stats_ws", ctypes.c_void_p), ("stats_ws_stride", ctypes.c_size_t),
                ("sqerr_partial", ctypes.c_void_p), ("sqerr_stride", ctypes.c_int64),
                ("chunks", ctypes.c_int64), ("chunk_streams", ctypes.c_void_p), ("stats_stream", ctypes.c_void_p),
                ("events", ctypes.c_void_p), ("n_events", ctypes.c_int64),
                ("groups", ctypes.c_int64), ("x_gstride", ctypes.c_int64), ("packed_gstride", ctypes.c_int64), ("embed_gstride", ctypes.c_int64),
                ("codes_gstride", ctypes.c_int64), ("stats_gstride", ctypes.c_int64), ("stats_ws_gstride", ctypes.c_size_t),
                ("sqerr_gstride", ctypes.c_int64), ("decode_out", ctypes.c_void_p), ("decode_ldo", ctypes.c_int64), ("decode_gstride", ctypes.c_int64)]


_CHAIN_EVENTS = {}


def _chain_events(device, main, n):
    """n reusable events that pair with `main` (vqhip_rvq_chain_forward re-records them on every call); -> (events, array of handles)"""
    evs = _lru(_CHAIN_EVENTS, (torch.device(device).index, main.cuda_stream), list)
    while len(evs) < n:
        e = torch.cuda.Event()
        e.record(main)                       # (torch creates the hipEvent_t lazily, at the first record)
        evs.append(e)
    return evs[:n], (ctypes.c_void_p * n)(*[e.cuda_event for e in evs[:n]])


@_on_device
def rvq_chain_forward(x: torch.Tensor, packed: torch.Tensor, embed: torch.Tensor, Q: int, *, row_mask=None, route_mode=0, row_chunks=1,
                      stats=None, stats_ws=None, sq_parts=None, stats_stream=None, groups=1, decode_out=None):
    """The residual loop (rvq.py:469-568) in ONE library call (vqhip_rvq_chain_forward): the Q chained screened searches -- or, with
    route_mode, the routed residuals + plain searches of a gradient step -- in row_chunks interleaved chunks, and every stage's EMA
    statistics (+ loss partials into sq_parts [Q, P]) on stats_stream.  Same results as rvq_forward_chained with a stage hook that
    calls ema_accumulate; the host enqueues one call instead of ~100 launches from Python.
    stats [Q, stride] zeroed, stats_ws from ema_workspaces(Q, N, C) (histograms zeroed).  The statistics stream is NOT joined here.
    -> dict(idx [..., Q], inputs (list of Q views), bufs [Q - 1, N, D], counts (per stage: per-chunk counters))
    groups = G > 1 (round 6): the G independent loops of a GroupedResidualVQ (rvq.py:634-724) as one launch set.  x [..., G D] -- group g
    owns the feature chunk g D .. (g + 1) D --, embed [G, C, D] (every group one codebook shared by its stages) or [G, Q, C, D], packed
    [G, P] / [G, Q, P], stats [G, Q, stride], stats_ws [G, Q, bytes], sq_parts [G, Q, P].
    -> idx [G, ..., Q], bufs [Q - 1, G, N, D], inputs None, counts per stage: counters [K, G]
    decode_out (fp32, x's shape): receives quantized_out = the sum of the chosen codes -- stages 0 .. Q - 2 summed on the statistics stream
    beside the last stage's search, the last stage added behind the loop (chain_decode_supported() says when this applies)."""
    _need_gpu(x, packed, embed, row_mask, stats, stats_ws, sq_parts)
    G = int(groups)
    if G > 1:
        assert x.shape[-1] % G == 0
        shared = embed.ndim == 3
        xk, N, Dall, ldx = as_rows(x)
        D = Dall // G
        assert embed.shape[0] == G and packed.shape[0] == G and (shared or (embed.shape[1] >= Q and packed.shape[1] >= Q))
    else:
        shared = embed.ndim == 2
        xk, N, D, ldx = as_rows(x)
    lead, dev = x.shape[:-1], x.device
    C = embed.shape[-2]
    assert embed.dtype == torch.float32 and embed.is_contiguous() and packed.is_contiguous() and embed.shape[-1] == D
    assert xk.dtype == torch.float32 or (route_mode and xk.dtype == torch.bfloat16), "chained stages: float32 rows"
    idx = torch.empty(*((G,) if G > 1 else ()), N, Q, dtype=torch.int64, device=dev)
    bufs = torch.empty(max(Q - 1, 1), *((G,) if G > 1 else ()), N, D, dtype=xk.dtype, device=dev)
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    main = torch.cuda.current_stream(dev)
    K = max(1, int(row_chunks))
    rpc = lib().vqhip_rvq_chain_chunk_rows(N, K)
    K = (N + rpc - 1) // rpc
    wss = lib().vqhip_rvq_chain_ws_stride(N, K)
    ws = torch.empty(Q * K * G, wss, dtype=torch.uint8, device=dev)
    codes = None
    if route_mode:
        codes = embed if xk.dtype == torch.float32 else embed.to(xk.dtype)      # (bf16 rows: the routing kernel gathers bf16 code rows)
    pfl = packed.element_size()
    st = _RvqChain(x=xk.data_ptr(), x_dtype=_dtype_code(xk), N=N, D=D, ldx=ldx, packed=packed.data_ptr(),
                   packed_qstride=0 if shared else packed.stride(-2) * pfl // 4, embed=embed.data_ptr(),
                   embed_qstride=0 if shared else embed.stride(-3), C=C, Q=Q, idx_out=idx.data_ptr(), inputs=bufs.data_ptr(),
                   row_mask=None if row_mask is None else row_mask.data_ptr(), workspace=ws.data_ptr(), workspace_bytes=ws.numel(),
                   route_mode=int(route_mode), codes=None if codes is None else codes.data_ptr(),
                   codes_qstride=0 if (codes is None or shared) else codes.stride(-3), chunks=K)
    if G > 1:
        st.groups, st.x_gstride = G, D
        st.packed_gstride, st.embed_gstride = packed.stride(0) * pfl // 4, embed.stride(0)
        st.codes_gstride = 0 if codes is None else codes.stride(0)
    keep = [codes]
    if stats is not None:
        assert stats.dtype == torch.float32 and stats.stride(-1) == 1 and stats.shape[-2] >= Q and stats.ndim == (3 if G > 1 else 2)
        assert stats_ws is not None and stats_ws.dtype == torch.uint8 and stats_ws.shape[-2] >= Q and stats_ws.stride(-1) == 1 and stats_ws.data_ptr() % 256 == 0
        st.stats, st.stats_stride, st.stats_ws, st.stats_ws_stride = stats.data_ptr(), stats.stride(-2), stats_ws.data_ptr(), stats_ws.stride(-2)
        if G > 1:
            assert stats.shape[0] == G and stats_ws.shape[0] == G and stats_ws.stride(0) % 256 == 0
            st.stats_gstride, st.stats_ws_gstride = stats.stride(0), stats_ws.stride(0)
        if sq_parts is not None:
            assert sq_parts.dtype == torch.float64 and sq_parts.shape[-2] >= Q and sq_parts.stride(-1) == 1
            st.sqerr_partial, st.sqerr_stride = sq_parts.data_ptr(), sq_parts.stride(-2)
            if G > 1:
                st.sqerr_gstride = sq_parts.stride(0)
        if stats_stream is not None:
            st.stats_stream = stats_stream.cuda_stream
    side = stats is not None and stats_stream is not None and stats_stream.cuda_stream != main.cuda_stream
    n_ev = Q * K + 1
    if decode_out is not None:
        assert side and Q >= 2 and row_mask is None and xk.dtype == torch.float32 and decode_out.dtype == torch.float32
        ok_, oN, oD, ldo = as_rows(decode_out)
        assert ok_.data_ptr() == decode_out.data_ptr() and oN == N and oD == D * G
        st.decode_out, st.decode_ldo, st.decode_gstride = decode_out.data_ptr(), ldo, D
        n_ev += 1
    if K > 1 or side:
        if K > 1:
            cs = _chain_streams(dev, main, K - 1)
            arr = (ctypes.c_void_p * (K - 1))(*[s_.cuda_stream for s_ in cs])
            st.chunk_streams = ctypes.cast(arr, ctypes.c_void_p)
            keep.append(arr)
        evs, earr = _chain_events(dev, main, n_ev)
        st.events, st.n_events = ctypes.cast(earr, ctypes.c_void_p), n_ev
        keep += [evs, earr]
    _check(lib().vqhip_rvq_chain_forward(ctypes.byref(st), _stream()), "vqhip_rvq_chain_forward")
    # keepalive: the statistics stream is not joined here and still reads the uint8 row mask made above (allocated on the CALLER's
    # stream: freed at return, the caching allocator would hand its memory to the caller's next allocation while those kernels run)
    if G > 1:
        wsi = ws.view(Q, K, G, wss)
        counts = [(wsi[q, :, :, 0:4].view(torch.int32)[..., 0], wsi[q, :, :, 4:8].view(torch.int32)[..., 0]) for q in range(Q)]
        return dict(idx=idx.view(G, *lead, Q), inputs=None, counts=counts, bufs=bufs, keepalive=(row_mask, codes, ws))
    wsi = ws.view(Q, K, wss)
    counts = [(wsi[q, :, 0:4].view(torch.int32)[:, 0], wsi[q, :, 4:8].view(torch.int32)[:, 0]) for q in range(Q)]
    inputs = [x] + [bufs[q].view(*lead, D) for q in range(Q - 1)]
    return dict(idx=idx.view(*lead, Q), inputs=inputs, counts=counts, bufs=bufs, keepalive=(row_mask, codes, ws))


def mask_fill_indices(idx: torch.Tensor, row_mask: torch.Tensor):
    """as the fused kernel: masked rows carry index -1 (decode contributes nothing)"""
    idx.masked_fill_(~row_mask.reshape(*idx.shape[:-1], 1).bool(), -1)


@_on_device
def rvq_forward_screened(x: torch.Tensor, packed: torch.Tensor, embed: torch.Tensor, Q: int, *, want_resid=False,
                         want_sqerr=False, row_mask=None, stage_hook=None, fill_masked=True):
    """The residual loop (rvq.py:469-568) as Q screened assignments: each stage's search runs on the fp16 MFMA pipe
    (csrc/vq_screen.hip) and writes the next stage's input x - q itself, so no N x D tensor op runs between stages.
    Same arguments as rvq_forward (+ stage_hook(q, stage_input, idx), called after stage q's launches);
    -> dict(idx [..., Q], inputs = the Q stage inputs (inputs[0] is x) | None, sqerr_partials [Q, P] | None)."""
    _need_gpu(x, packed, embed, row_mask)
    shared = embed.ndim == 2
    lead, D, dev = x.shape[:-1], x.shape[-1], x.device
    idx = torch.empty(*lead, Q, dtype=torch.int64, device=dev)
    nbuf = (Q - 1) if want_resid else min(Q - 1, 2)
    bufs = torch.empty(max(nbuf, 1), *lead, D, dtype=x.dtype, device=dev)
    inputs, parts, cur = [], [], x
    for q in range(Q):
        nxt = None if q + 1 == Q else bufs[q if want_resid else q % 2]
        r = assign(cur, packed if shared else packed[q], embed if shared else embed[q], want_q=False, want_sqerr=want_sqerr,
                   row_mask=row_mask, resid_out=nxt)
        idx[..., q] = r["idx"]
        if want_sqerr:
            parts.append(r["sqerr_partials"][: r["nblk"]])
        inputs.append(cur)
        if stage_hook is not None:      # stage q's input and indices are final (in stream order): the caller's per-stage work
            stage_hook(q, cur, idx)
        cur = nxt
    if row_mask is not None and fill_masked:
        mask_fill_indices(idx, row_mask)
    return dict(idx=idx, resid=None, inputs=inputs if want_resid else None,
                sqerr_partials=torch.stack(parts) if want_sqerr else None)


@_on_device
def rvq_forward(x: torch.Tensor, packed: torch.Tensor, embed: torch.Tensor, Q: int, *, want_resid=False,
                want_sqerr=False, row_mask=None):
    """Fused residual loop.  embed [C, D] (shared by all stages; packed = pack_codebook(embed)) or [Q, C, D]
    (packed = [Q, packed_floats]).  -> dict(idx [..., Q], resid [..., Q, D] | None, sqerr_partials [Q, 4*nblk] | None)."""
    _need_gpu(x, packed, embed, row_mask)
    xk, N, D, ldx = as_rows(x)
    assert embed.dtype == torch.float32 and embed.is_contiguous() and embed.shape[-1] == D
    if embed.ndim == 2:
        C, eq, pq = embed.shape[0], 0, 0
    else:
        assert embed.shape[0] == Q and packed.ndim == 2 and packed.shape[0] == Q and packed.is_contiguous()
        C, eq, pq = embed.shape[1], embed.shape[1] * D, packed.shape[1]
    dev, lead = x.device, x.shape[:-1]
    idx = torch.empty(*lead, Q, dtype=torch.int64, device=dev)
    resid = torch.empty(*lead, Q, D, dtype=x.dtype, device=dev) if want_resid else None
    nblk = lib().vqhip_assign_blocks(N)
    partials = torch.zeros(Q, 4 * max(nblk, 1), dtype=torch.float64, device=dev) if want_sqerr else None
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    if N > 0:
        _check(lib().vqhip_rvq_forward(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(packed), pq, _ptr(embed), eq, C, Q,
                                       _ptr(idx), _ptr(resid), _ptr(partials), _ptr(row_mask), _stream()),
               "vqhip_rvq_forward")
    return dict(idx=idx, resid=resid, sqerr_partials=partials, nblk=nblk)


STRAIGHT_THROUGH, ROTATION = 1, 2


@_on_device
def route_fwd(x: torch.Tensor, q: torch.Tensor, mode: int) -> torch.Tensor:
    """forward value of straight-through (mode 1) / the rotation trick (mode 2), rows = last dim."""
    _need_gpu(x, q)
    assert x.dtype == q.dtype and x.shape == q.shape
    xk, N, D, ldx = as_rows(x)
    qk, _, _, ldq = as_rows(q)
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if N > 0:
        _check(lib().vqhip_route_fwd(_ptr(xk), _ptr(qk), _dtype_code(xk), N, D, ldx, ldq, _ptr(out), D, mode, _stream()),
               "vqhip_route_fwd")
    return out


@_on_device
def route_fwd_gather(x: torch.Tensor, codes: torch.Tensor, idx: torch.Tensor, mode: int) -> torch.Tensor:
    """route_fwd with q = codes[idx] gathered inside the kernel (codes [C, D] in x's dtype): no [N, D] q tensor"""
    _need_gpu(x, codes, idx)
    assert codes.dtype == x.dtype and codes.is_contiguous() and codes.ndim == 2 and idx.dtype == torch.int64 and idx.is_contiguous()
    xk, N, D, ldx = as_rows(x)
    assert idx.numel() == N and codes.shape[1] == D
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if N > 0:
        _check(lib().vqhip_route_fwd_gather(_ptr(xk), _ptr(codes), _ptr(idx), 1, _dtype_code(xk), N, D, ldx, _ptr(out), D, mode, _stream()),
               "vqhip_route_fwd_gather")
    return out


@_on_device
def route_bwd_gather(x: torch.Tensor, codes: torch.Tensor, idx: torch.Tensor, g_out, loss_coef, row_mask, mode: int,
                     masked_rows: int = 0) -> torch.Tensor:
    """route_bwd with q = codes[idx] gathered inside the kernel"""
    _need_gpu(x, codes, idx, g_out, loss_coef, row_mask)
    xk, N, D, ldx = as_rows(x)
    gk, ldg = None, 0
    if g_out is not None and mode != 0:
        gk, _, _, ldg = as_rows(g_out.to(x.dtype))
    if loss_coef is not None:
        loss_coef = loss_coef.to(torch.float32).reshape(()).contiguous()
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    gx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if N > 0:
        _check(lib().vqhip_route_bwd_gather(_ptr(xk), _ptr(codes), _ptr(idx), 1, _ptr(gk), _dtype_code(xk), N, D, ldx, ldg, _ptr(loss_coef),
                                            _ptr(row_mask), mode if gk is not None else 0, int(masked_rows), _ptr(gx), D, _stream()),
               "vqhip_route_bwd_gather")
    return gx


@_on_device
def mask_fill_rows(q, x, row_mask: torch.Tensor, idx=None, *, zeros=False):
    """In place: the rows with row_mask == 0 of q [..., D] take x's rows (zeros=True: zeros) and their entries of idx [...] become -1
    (the reference's two torch.where of a masked batch, vqp.py:1386-1394); touches the padding rows only."""
    _need_gpu(q, x, row_mask, idx)
    row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    N = row_mask.numel()
    qk = xk = None
    D = ldq = ldx = 1
    dt = F32
    if q is not None:
        qk, Nq, D, ldq = as_rows(q)
        assert Nq == N and qk.data_ptr() == q.data_ptr(), "mask_fill_rows: q must be row-addressable in place"
        dt = _dtype_code(qk)
        if not zeros:
            xk, Nx, Dx, ldx = as_rows(x)
            assert Nx == N and Dx == D and xk.dtype == qk.dtype
    if idx is not None:
        assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == N
    if N > 0:
        _check(lib().vqhip_mask_fill_rows(_ptr(qk), _ptr(xk), dt, N, D, ldq, ldx, _ptr(row_mask), _ptr(idx), 1, int(bool(zeros)), _stream()),
               "vqhip_mask_fill_rows")


@_on_device
def route_bwd(x: torch.Tensor, q: torch.Tensor, g_out, loss_coef, row_mask, mode: int, masked_rows: int = 0) -> torch.Tensor:
    """grad wrt x of (routed output, commit-loss sum); g_out may be None (mode 0), loss_coef a 0-dim fp32 device tensor or None."""
    _need_gpu(x, q, g_out, loss_coef, row_mask)
    xk, N, D, ldx = as_rows(x)
    qk, _, _, ldq = as_rows(q)
    gk, ldg = None, 0
    if g_out is not None and mode != 0:
        gk, _, _, ldg = as_rows(g_out.to(x.dtype))
    if loss_coef is not None:
        loss_coef = loss_coef.to(torch.float32).reshape(()).contiguous()
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    gx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if N > 0:
        _check(lib().vqhip_route_bwd(_ptr(xk), _ptr(qk), _ptr(gk), _dtype_code(xk), N, D, ldx, ldq, ldg, _ptr(loss_coef),
                                     _ptr(row_mask), mode if gk is not None else 0, int(masked_rows), _ptr(gx), D, _stream()), "vqhip_route_bwd")
    return gx


@_on_device
def rvq_route(x: torch.Tensor, embed: torch.Tensor, idx: torch.Tensor, Q: int, mode: int, *, g_out=None, loss_coef=None,
              row_mask=None, backward=False, resid_routed=False, loss_only=False, out=None) -> torch.Tensor:
    """The residual loop's routed output (backward=False) or the gradient wrt x (backward=True) in ONE kernel that keeps the
    residual row in registers (csrc: vq_rvq_route_kernel; rvq.py:469-568 with quant_grad_frac = 0).
    x [..., D]; embed fp32 [Q', C, D] or [C, D] (shared); idx int64 [..., Q'] (first Q columns are used); mode 0 / STRAIGHT_THROUGH /
    ROTATION; loss_coef: [Q] fp32 device tensor, d loss / d (sum of squared errors of stage q).  resid_routed: the residuals are
    re-derived as r - route(r, c) (what rvq.py:524 subtracts when the layers returned routed values) instead of r - c.
    out: where the result goes -- x's shape and dtype, rows at a uniform stride (a feature chunk of a wider tensor: the groups of a
    GroupedResidualVQ write their columns of one output)."""
    _need_gpu(x, embed, idx, g_out, loss_coef, row_mask)
    xk, N, D, ldx = as_rows(x)
    assert idx.dtype == torch.int64 and idx.is_contiguous() and embed.dtype == torch.float32 and embed.is_contiguous()
    qs = idx.shape[-1]
    assert 1 <= Q <= qs and idx.numel() == N * qs
    if embed.ndim == 2:
        C, qstride = embed.shape[0], 0
    else:
        assert embed.shape[0] >= Q
        C, qstride = embed.shape[1], embed.shape[1] * embed.shape[2]
    assert embed.shape[-1] == D
    gk, ldg = None, 0
    if backward and mode != 0 and not loss_only:       # loss_only: no upstream gradient for the output (treated as zero)
        gk, _, _, ldg = as_rows(g_out.to(x.dtype))
    if loss_coef is not None:
        loss_coef = loss_coef.to(torch.float32).reshape(-1).contiguous()
        assert loss_coef.numel() >= Q
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
        assert row_mask.numel() == N
    ldo = D
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    else:
        ok_, oN, oD, ldo = as_rows(out)
        assert out.shape == x.shape and out.dtype == x.dtype and ok_.data_ptr() == out.data_ptr() and oN == N and oD == D, "rvq_route: out must be rows at a uniform stride"
    if N > 0:
        _check(lib().vqhip_rvq_route(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(embed), qstride, C, _ptr(idx), qs, Q, mode,
                                     1 if resid_routed else 0, _ptr(gk), ldg, _ptr(loss_coef), _ptr(row_mask), 1 if backward else 0, _ptr(out), ldo, _stream()),
               "vqhip_rvq_route")
    return out


@_on_device
def reduce_partials(partials: torch.Tensor, n: int, scale: float, out: torch.Tensor | None = None) -> torch.Tensor:
    _need_gpu(partials)
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=partials.device)
    _check(lib().vqhip_reduce_partials(_ptr(partials), n, float(scale), _ptr(out), _stream()), "vqhip_reduce_partials")
    return out


@_on_device
def reduce_partials_rows(partials: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """partials [R, P] float64 -> [R] float32: one launch for the per-stage losses of a residual VQ"""
    _need_gpu(partials)
    assert partials.dtype == torch.float64 and partials.ndim == 2 and partials.stride(1) == 1
    out = torch.empty(partials.shape[0], dtype=torch.float32, device=partials.device)
    _check(lib().vqhip_reduce_partials_rows(_ptr(partials), partials.shape[0], partials.shape[1], partials.stride(0), float(scale),
                                            _ptr(out), _stream()), "vqhip_reduce_partials_rows")
    return out


@_on_device
def ema_fold_many(cluster_size, embed_avg, embed, stats, *, decay, eps, cosine=False, do_update_ema=True, denom_ws=None):
    """Q successive folds of one (shared) codebook in one call: stats [Q, stride >= C D + C] float32, each row embed_sum [C, D]
    followed by count [C]; in place on cluster_size [C], embed_avg [C, D], embed [C, D]."""
    _need_gpu(cluster_size, embed_avg, embed, stats)
    C, D = embed.shape
    for t in (cluster_size, embed_avg, embed):
        assert t.is_contiguous() and t.dtype == torch.float32
    assert stats.dtype == torch.float32 and stats.ndim == 2 and stats.stride(1) == 1 and stats.shape[1] >= C * D + C
    if denom_ws is None and do_update_ema:
        denom_ws = torch.empty(C, dtype=torch.float32, device=embed.device)
    omd = float(torch.tensor(1.0 - decay, dtype=torch.float64).to(torch.float32))
    _check(lib().vqhip_ema_fold_many(_ptr(cluster_size), _ptr(embed_avg), _ptr(embed), _ptr(stats), stats.shape[0], stats.stride(0),
                                     C, D, omd, float(eps), int(cosine), int(do_update_ema), _ptr(denom_ws), _stream()), "vqhip_ema_fold_many")


@_on_device
def stats_sqerr_supported(x: torch.Tensor, cosine=False) -> bool:
    """can the statistics pass also sum the commitment loss' squared error for these rows (vqhip_ema_accumulate_sqerr)?"""
    if cosine or not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16) or os.environ.get("VQHIP_STATS_SQERR", "1") == "0":
        return False
    xk, N, D, ldx = as_rows(x)
    es = xk.element_size()
    return D % 4 == 0 and D <= WIDE_MAX_DIM and xk.data_ptr() % (4 * es) == 0 and (ldx * es) % (4 * es) == 0


@_on_device
def ema_accumulate(x: torch.Tensor, idx: torch.Tensor, C: int, *, cosine=False, rnorm=None, row_mask=None,
                   count=None, embed_sum=None, idx_stride=1, idx_offset=0, sqerr_from=None, sqerr_out=None, ws=None):
    """Accumulates into (count [C], embed_sum [C, D]); allocates zeroed ones if not given.
    sqerr_from = (packed, embed): also returns the squared-error partials of the commitment loss, summed by the same pass
    (-> count, embed_sum, partials [P] float64); requires stats_sqerr_supported(x)."""
    _need_gpu(x, idx, rnorm, row_mask)
    xk, N, D, ldx = as_rows(x)
    dev = x.device
    if count is None:
        count = torch.zeros(C, dtype=torch.float32, device=dev)
    if embed_sum is None:
        embed_sum = torch.zeros(C, D, dtype=torch.float32, device=dev)
    assert idx.dtype == torch.int64
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    if N > 0:
        nbytes = lib().vqhip_ema_workspace_bytes(N, C)
        idx_ptr = ctypes.c_void_p(idx.data_ptr() + 8 * idx_offset)     # e.g. column q of an [N, Q] index tensor
        if ws is not None:
            # a workspace of the caller's whose histogram (the first C ints) it has zeroed already, e.g. for all stages of a residual
            # VQ at once (ema_workspaces): no memset launch queued on the statistics stream
            assert not cosine and ws.dtype == torch.uint8 and ws.is_contiguous() and ws.numel() >= nbytes and ws.data_ptr() % 256 == 0
            partials = None
            pk = em = None
            if sqerr_from is not None:
                pk, em = sqerr_from
                npart = lib().vqhip_ema_sqerr_partials(N, C)
                partials = sqerr_out if sqerr_out is not None else torch.empty(npart, dtype=torch.float64, device=dev)
                assert partials.dtype == torch.float64 and partials.is_contiguous() and partials.numel() == npart
            _check(lib().vqhip_ema_accumulate_prezeroed(_ptr(xk), _dtype_code(xk), N, D, ldx, idx_ptr, idx_stride, _ptr(row_mask), C,
                                                        _ptr(count), _ptr(embed_sum), _ptr(ws), ws.numel(), _ptr(pk), _ptr(em),
                                                        _ptr(partials), _stream()), "vqhip_ema_accumulate_prezeroed")
            return (count, embed_sum, partials) if sqerr_from is not None else (count, embed_sum)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)     # caching allocator: 512-byte aligned
        if sqerr_from is not None:
            packed, embed = sqerr_from
            assert not cosine and embed.dtype == torch.float32 and embed.is_contiguous() and tuple(embed.shape) == (C, D)
            npart = lib().vqhip_ema_sqerr_partials(N, C)
            if sqerr_out is not None:          # a row of a caller-owned [Q, P] buffer (one batched reduction afterwards)
                assert sqerr_out.dtype == torch.float64 and sqerr_out.is_contiguous() and sqerr_out.numel() == npart
                partials = sqerr_out
            else:
                partials = torch.empty(npart, dtype=torch.float64, device=dev)
            _check(lib().vqhip_ema_accumulate_sqerr(_ptr(xk), _dtype_code(xk), N, D, ldx, idx_ptr, idx_stride, _ptr(row_mask), C,
                                                    _ptr(count), _ptr(embed_sum), _ptr(ws), nbytes, _ptr(packed), _ptr(embed),
                                                    _ptr(partials), _stream()), "vqhip_ema_accumulate_sqerr")
            return count, embed_sum, partials
        _check(lib().vqhip_ema_accumulate(_ptr(xk), _dtype_code(xk), N, D, ldx, idx_ptr, idx_stride, _ptr(rnorm),
                                          COSINE if cosine else EUCLID, _ptr(row_mask), C, _ptr(count),
                                          _ptr(embed_sum), _ptr(ws), nbytes, _stream()), "vqhip_ema_accumulate")
    if sqerr_from is not None:
        return count, embed_sum, torch.zeros(1, dtype=torch.float64, device=dev)
    return count, embed_sum


class _Step(ctypes.Structure):           # vqhip_vq_step_t (include/vqhip.h)
    _fields_ = [("x", ctypes.c_void_p), ("x_dtype", ctypes.c_int64), ("N", ctypes.c_int64), ("D", ctypes.c_int64), ("ldx", ctypes.c_int64),
                ("embed", ctypes.c_void_p), ("embed_avg", ctypes.c_void_p), ("cluster_size", ctypes.c_void_p), ("C", ctypes.c_int64),
                ("idx_out", ctypes.c_void_p), ("q_out", ctypes.c_void_p), ("ldq", ctypes.c_int64),
                ("stats", ctypes.c_void_p), ("loss_out", ctypes.c_void_p), ("loss_scale", ctypes.c_double),
                ("packed", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("one_minus_decay", ctypes.c_double), ("eps", ctypes.c_double), ("fold", ctypes.c_int64),
                ("ev_search_begin", ctypes.c_void_p), ("ev_search_end", ctypes.c_void_p), ("metric", ctypes.c_int64),
                ("row_mask", ctypes.c_void_p), ("chunks", ctypes.c_int64), ("side_stream", ctypes.c_void_p), ("events", ctypes.c_void_p * 4)]


step_event_hook = None   # bench.py: callable -> (begin, end) torch.cuda.Event pair (already recorded once, so that their handles exist),
                         # recorded by the library around the search of the next fused step


_STEP_SIDE = {}


def _step_side(device, main):
    """the side stream + events of the fused step's row pipeline that pair with `main` (created once per caller stream)"""
    def make():
        evs = [torch.cuda.Event() for _ in range(4)]
        for e in evs:
            e.record(main)          # (torch creates the hipEvent_t lazily, at the first record)
        return (torch.cuda.Stream(device=device), evs)
    return _lru(_STEP_SIDE, (torch.device(device).index, main.cuda_stream), make)


def step_chunks(N: int) -> int:
    """row chunks of the fused train step's pipeline (include/vqhip.h, vqhip_vq_step_t.chunks): VQHIP_STEP_CHUNKS, default 1 -- the
    pipeline is built, tested and measured, and it LOSES at cfg 2 (0.845 -> 0.952 ms with 2 chunks, 1.03 with 4; DESIGN 5.1,
    profiles/r5_step_pipeline): the screening kernels fill every SIMD's register file, so the side stream's kernels only start when a
    screening workgroup retires (its single-workgroup scan kernel waited 226 us for a slot).  Batches of at least 2^19 rows, each
    chunk at least 2^17 rows (the persistent screening kernel's floor); 1 while a HIP graph is being captured."""
    k = int(os.environ.get("VQHIP_STEP_CHUNKS", "1"))
    if N < (1 << 19) or torch.cuda.is_current_stream_capturing():
        return 1
    return max(1, min(k, 4, N // (1 << 17)))


def vq_step_supported(x: torch.Tensor, C: int) -> bool:
    """can this training forward run as ONE fused call (vqhip_vq_train_step)?  VQHIP_FUSED_STEP=0 keeps the separate calls."""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and screening_enabled()):
        return False
    if os.environ.get("VQHIP_FUSED_STEP", "1") == "0" or os.environ.get("VQHIP_SCREEN_VERIFY", "0") == "1" or screen_debug:
        return False
    xk, N, D, ldx = as_rows(x)
    es = xk.element_size()
    return bool(N > 0 and lib().vqhip_vq_step_supported(_dtype_code(xk), N, D, C) and xk.data_ptr() % 16 == 0 and (ldx * es) % 16 == 0)


_SCRATCH = {}


def scratch(kind: str, nbytes: int, device) -> torch.Tensor:
    """A persistent scratch buffer for a library call: one per (device, CURRENT STREAM, kind, size), so consecutive calls on a stream
    reuse it in stream order (safe: the library only enqueues) and concurrent streams -- GroupedResidualVQ's groups -- never share
    one.  Replaces a 26 - 31 MiB torch.empty per forward on the training hot path.  At most 16 buffers are kept
    (least recently used first out); contents are only valid until the next call that asks for the same key."""
    dev = torch.device(device)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, kind, int(nbytes))
    t = _SCRATCH.pop(key, None)
    if t is None:
        t = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
        while len(_SCRATCH) >= 16:
            _SCRATCH.pop(next(iter(_SCRATCH)))
    _SCRATCH[key] = t                      # (re-inserted: dicts keep insertion order, so the first key is the least recently used)
    return t


@_on_device
def vq_train_step(x: torch.Tensor, embed, embed_avg, cluster_size, *, decay, eps, want_q=True, q_out=None, loss_scale=None, fold=True,
                  cosine=False, row_mask=None, reuse_scratch=False):
    """One training forward of an EMA codebook (Euclidean, or cosine=True on rows already unit-norm: l2norm_rows) in one library call (vqhip_vq_train_step; reference: vqp.py:673-800 under
    VectorQuantize.forward :1176).  embed / embed_avg / cluster_size: [C, D], [C, D], [C] fp32, updated in place when fold.
    -> dict(q, idx, count [C], embed_sum [C, D] (views of one [C D + C] buffer: one all-reduce), loss (0-dim fp32 or None))"""
    _need_gpu(x, embed, embed_avg, cluster_size, row_mask)
    xk, N, D, ldx = as_rows(x)
    C = embed.shape[0]
    dev = x.device
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
        assert row_mask.numel() == N
    for t in (embed, embed_avg, cluster_size):
        assert t.is_contiguous() and t.dtype == torch.float32
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    q = None
    if want_q:
        q = q_out if q_out is not None else torch.empty(N, D, dtype=xk.dtype, device=dev)
        assert q.dtype == xk.dtype and q.is_contiguous() and q.numel() == N * D
    stats = torch.empty(C * D + C, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev) if loss_scale is not None else None
    nws = lib().vqhip_vq_step_workspace_bytes(N, C)
    if reuse_scratch and not torch.cuda.is_current_stream_capturing():
        # the packed codebook and the workspace live only inside this call (n_exact / n_pair below are views of the workspace: read
        # them before the next step on this stream)
        packed = scratch("step.packed", lib().vqhip_packed_bytes(C, D), dev)
        ws = scratch("step.ws", nws, dev)
    else:
        packed = torch.empty(lib().vqhip_packed_bytes(C, D), dtype=torch.uint8, device=dev)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    omd = float(torch.tensor(1.0 - decay, dtype=torch.float64).to(torch.float32))
    st = _Step(x=xk.data_ptr(), x_dtype=_dtype_code(xk), N=N, D=D, ldx=ldx, embed=embed.data_ptr(), embed_avg=embed_avg.data_ptr(),
               cluster_size=cluster_size.data_ptr(), C=C, idx_out=idx.data_ptr(), q_out=None if q is None else q.data_ptr(), ldq=D,
               stats=stats.data_ptr(), loss_out=None if loss is None else loss.data_ptr(), loss_scale=float(loss_scale or 0.0),
               packed=packed.data_ptr(), workspace=ws.data_ptr(), workspace_bytes=nws, one_minus_decay=omd, eps=float(eps), fold=int(bool(fold)),
               metric=COSINE_PRENORM if cosine else EUCLID, row_mask=None if row_mask is None else row_mask.data_ptr())
    if step_event_hook is not None:
        e0, e1 = step_event_hook()
        st.ev_search_begin, st.ev_search_end = e0.cuda_event, e1.cuda_event
    K = step_chunks(N)
    if K > 1:       # row pipeline: the statistics of chunk k on a side stream beside the search of chunk k + 1
        side, evs = _step_side(dev, torch.cuda.current_stream(dev))
        st.chunks, st.side_stream = K, side.cuda_stream
        for k in range(K):
            st.events[k] = evs[k].cuda_event
    _check(lib().vqhip_vq_train_step(ctypes.byref(st), _stream()), "vqhip_vq_train_step")
    # per chunk: [0] rows of the exact sweep, [1] rows decided between two candidates (device-side counters at the head of the chunk's
    # screening workspace, include/vqhip.h)
    rpc = lib().vqhip_vq_step_chunk_rows(N, K)
    offs, o = [], 0
    for r0 in range(0, N, rpc):
        n_k = min(rpc, N - r0)
        offs.append(o)
        o += (lib().vqhip_screen_workspace_bytes(n_k) + 255) // 256 * 256 + (lib().vqhip_ema_workspace_bytes(n_k, C) + 255) // 256 * 256
    hdr = ws[:16].view(torch.int32) if len(offs) == 1 else torch.stack([ws[o: o + 16].view(torch.int32) for o in offs]).sum(0)
    if reuse_scratch and len(offs) == 1:
        hdr = hdr.clone()            # (the persistent workspace is overwritten by the next step on this stream: hand out a copy of the counters)
    return dict(q=None if q is None else q.reshape(x.shape), idx=idx.reshape(x.shape[:-1]), stats=stats,
                embed_sum=stats[: C * D].view(C, D), count=stats[C * D:], loss=loss, n_exact=hdr[:1], n_pair=hdr[1:2])


@_on_device
def ema_accumulate_batched(xs: torch.Tensor, idx: torch.Tensor, C: int, stats: torch.Tensor, *, row_mask=None, sqerr_from=None):
    """The statistics of H heads in one set of launches: xs [H, ..., D] (uniform strides, as for assign_batched), idx [H, ...], stats
    [H, stride >= C D + C] float32, ZEROED by the caller (embed_sum || count per head: one buffer, one all-reduce under data
    parallelism).  sqerr_from = (packed [H, P], embed [H, C, D]): also the squared-error partials of the commitment loss [H, P']."""
    _need_gpu(xs, idx, stats, row_mask)
    H = xs.shape[0]
    xk, N, D, ldx = as_rows(xs[0])
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == H * N
    assert stats.dtype == torch.float32 and stats.ndim == 2 and stats.shape[0] == H and stats.stride(1) == 1 and stats.shape[1] >= C * D + C
    dev = xs.device
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    wss = lib().vqhip_ema_batched_ws_stride(N, C)
    ws = torch.empty(H * wss, dtype=torch.uint8, device=dev)
    pk = em = parts = None
    if sqerr_from is not None:
        pk, em = sqerr_from
        assert em.dtype == torch.float32 and em.is_contiguous() and tuple(em.shape) == (H, C, D) and pk.is_contiguous()
        parts = torch.empty(H, lib().vqhip_ema_sqerr_partials(N, C), dtype=torch.float64, device=dev)
    _check(lib().vqhip_ema_accumulate_batched(_ptr(xk), _dtype_code(xk), H, N, D, ldx, xs.stride(0), _ptr(idx), _ptr(row_mask), C,
                                              _ptr(stats), stats.stride(0), _ptr(ws), H * wss, _ptr(pk), _ptr(em), _ptr(parts), _stream()),
           "vqhip_ema_accumulate_batched")
    return parts


@_on_device
def ema_finalize_batched(cluster_size, embed_avg, embed, stats, *, decay, eps, cosine=False, do_update_ema=True):
    """ema_finalize for the H codebooks of a multi-head module in three launches: cluster_size [H, C], embed_avg / embed [H, C, D] (the
    module buffers, in place), stats [H, stride] as filled by ema_accumulate_batched."""
    _need_gpu(cluster_size, embed_avg, embed, stats)
    H, C, D = embed.shape
    for t in (cluster_size, embed_avg, embed):
        assert t.is_contiguous() and t.dtype == torch.float32
    denom = torch.empty(H, C, dtype=torch.float32, device=embed.device) if do_update_ema else None
    omd = float(torch.tensor(1.0 - decay, dtype=torch.float64).to(torch.float32))
    _check(lib().vqhip_ema_finalize_batched(_ptr(cluster_size), _ptr(embed_avg), _ptr(embed), _ptr(stats), stats.stride(0), H, C, D, omd,
                                            float(eps), int(cosine), int(do_update_ema), _ptr(denom), _stream()), "vqhip_ema_finalize_batched")


def pointer_table(tensors) -> torch.Tensor:
    """device int64 tensor of the tensors' data pointers (row-major over the nesting): the argument of vqhip_ema_finalize_table"""
    flat = [t.data_ptr() for row in tensors for t in row]
    dev = tensors[0][0].device
    return torch.tensor(flat, dtype=torch.int64).to(dev, non_blocking=False).view(len(tensors), -1)


@_on_device
def ema_finalize_table(table: torch.Tensor, stats: torch.Tensor, C: int, D: int, *, decay, eps, cosine=False, do_update_ema=True):
    """The folds of H codebooks kept in separate buffers (the layers of a grouped / residual VQ) in three launches: table [H, 3] from
    pointer_table([(cluster_size [C], embed_avg [C, D], embed [C, D]), ...]), stats [H, stride] = embed_sum || count per head."""
    _need_gpu(table, stats)
    H = table.shape[0]
    assert table.dtype == torch.int64 and table.is_contiguous() and table.shape[1] == 3
    assert stats.dtype == torch.float32 and stats.ndim == 2 and stats.shape[0] == H and stats.stride(1) == 1 and stats.shape[1] >= C * D + C
    denom = torch.empty(H, C, dtype=torch.float32, device=stats.device) if do_update_ema else None
    omd = float(torch.tensor(1.0 - decay, dtype=torch.float64).to(torch.float32))
    _check(lib().vqhip_ema_finalize_table(_ptr(table), _ptr(stats), stats.stride(0), H, C, D, omd, float(eps), int(cosine), int(do_update_ema),
                                          _ptr(denom), _stream()), "vqhip_ema_finalize_table")


@_on_device
def ema_accumulate_stages(inputs: torch.Tensor, idx: torch.Tensor, stage0: int, C: int, stats: torch.Tensor, ws: torch.Tensor, *,
                          row_mask=None, sqerr_from=None, sqerr_out=None):
    """The statistics of the stages stage0 .. stage0 + S - 1 of a residual VQ in one set of launches (vqhip_ema_accumulate_stages):
    inputs [S, ..., D] contiguous (the materialised stage inputs), idx [..., Q] int64 (stage s reads column stage0 + s), stats
    [S, stride >= C D + C] zeroed by the caller, ws [S, bytes] from ema_workspaces (histograms zeroed).
    sqerr_from = (packed, embed): shared codebook (packed 1-D, embed [C, D]) or per stage (packed [S, P], embed [S, C, D]);
    sqerr_out [S, P'] float64 receives the loss partials."""
    _need_gpu(inputs, idx, stats, ws, row_mask)
    S = inputs.shape[0]
    xk, N, D, ldx = as_rows(inputs[0])
    Q = idx.shape[-1]
    assert inputs.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == N * Q and stage0 + S <= Q
    assert stats.dtype == torch.float32 and stats.ndim == 2 and stats.shape[0] == S and stats.stride(1) == 1
    assert ws.dtype == torch.uint8 and ws.ndim == 2 and ws.shape[0] == S and ws.stride(1) == 1
    assert ws.stride(0) == lib().vqhip_ema_batched_ws_stride(N, C) and ws.data_ptr() % 256 == 0
    if row_mask is not None:
        row_mask = row_mask.reshape(-1).to(torch.uint8).contiguous()
    pk = em = None
    pks = ems = sqs = 0
    if sqerr_from is not None:
        pk, em = sqerr_from
        assert em.dtype == torch.float32 and em.is_contiguous() and pk.is_contiguous()
        if em.ndim == 3:
            assert em.shape[0] == S and pk.shape[0] == S
            ems, pks = em.stride(0), pk.stride(0) * pk.element_size() // 4
        assert sqerr_out.dtype == torch.float64 and sqerr_out.shape[0] == S and sqerr_out.stride(1) == 1
        sqs = sqerr_out.stride(0)
    _check(lib().vqhip_ema_accumulate_stages(_ptr(xk), _dtype_code(xk), S, N, D, ldx, inputs.stride(0), ctypes.c_void_p(idx.data_ptr() + 8 * stage0),
                                             Q, _ptr(row_mask), C, _ptr(stats), stats.stride(0), _ptr(ws), S * ws.stride(0), 1,
                                             _ptr(pk), pks, _ptr(em), ems, _ptr(sqerr_out) if sqerr_from is not None else None, sqs, _stream()),
           "vqhip_ema_accumulate_stages")


def ema_workspaces(Q: int, N: int, C: int, device) -> torch.Tensor:
    """[Q, bytes] statistics workspaces with their histograms zeroed in ONE launch (ema_accumulate(ws=...))"""
    nbytes = (lib().vqhip_ema_workspace_bytes(N, C) + 255) // 256 * 256
    ws = torch.empty(Q, nbytes, dtype=torch.uint8, device=device)
    ws[:, : C * 4].zero_()
    return ws


@_on_device
def ema_finalize(cluster_size, embed_avg, embed, count, embed_sum, *, decay, eps, cosine=False, weight=None,
                 do_lerp=True, do_update_ema=True, denom_ws=None):
    """In place on cluster_size [C], embed_avg [C, D], embed [C, D] (2-D views of the module buffers)."""
    _need_gpu(cluster_size, embed_avg, embed)
    C, D = embed.shape
    for t in (cluster_size, embed_avg, embed):
        assert t.is_contiguous() and t.dtype == torch.float32
    if denom_ws is None and do_update_ema:
        denom_ws = torch.empty(C, dtype=torch.float32, device=embed.device)
    # (1 - decay) is formed in Python double and handed to ATen as a scalar, which casts to fp32
    omd = float(torch.tensor(1.0 - decay, dtype=torch.float64).to(torch.float32))
    _check(lib().vqhip_ema_finalize(_ptr(cluster_size), _ptr(embed_avg), _ptr(embed), _ptr(count), _ptr(embed_sum),
                                    _ptr(weight), C, D, omd, float(eps), int(cosine), int(do_lerp),
                                    int(do_update_ema), _ptr(denom_ws), _stream()), "vqhip_ema_finalize")


@_on_device
def ema_renormalize_shard(cluster_size, embed_avg, embed, total_cluster_size, C_total, *, eps, cosine=False):
    """update_ema on one shard [C_local, D] of a codebook partitioned over ranks; total_cluster_size: 0-dim fp32 device tensor holding
    sum(cluster_size) over ALL shards (vqp.py:152-154, 576-584)."""
    _need_gpu(cluster_size, embed_avg, embed, total_cluster_size)
    C, D = embed.shape
    for t in (cluster_size, embed_avg, embed):
        assert t.is_contiguous() and t.dtype == torch.float32
    total = total_cluster_size.to(torch.float32).reshape(()).contiguous()
    ws = torch.empty(C, dtype=torch.float32, device=embed.device)
    _check(lib().vqhip_ema_renormalize_shard(_ptr(cluster_size), _ptr(embed_avg), _ptr(embed), C, D, float(eps), _ptr(total), int(C_total),
                                             int(cosine), _ptr(ws), _stream()), "vqhip_ema_renormalize_shard")


@_on_device
def assign_rowwise(x: torch.Tensor, codes: torch.Tensor, cosine=False) -> torch.Tensor:
    """x [..., D], codes [..., C, D] (one codebook per row, QINCo) -> idx [...]: nearest code of each row's own codebook
    (F.pairwise_distance arithmetic incl. its eps; cosine: dot products of unit-norm rows and codes)."""
    _need_gpu(x, codes)
    lead, D, C = x.shape[:-1], x.shape[-1], codes.shape[-2]
    assert codes.shape[:-2] == lead and codes.shape[-1] == D
    xf = x.detach().reshape(-1, D).float().contiguous()
    cf = codes.detach().reshape(-1, C, D).float().contiguous()
    idx = torch.empty(xf.shape[0], dtype=torch.int64, device=x.device)
    _check(lib().vqhip_assign_rowwise(_ptr(xf), xf.shape[0], D, D, _ptr(cf), C, COSINE_PRENORM if cosine else EUCLID, _ptr(idx), _stream()),
           "vqhip_assign_rowwise")
    return idx.reshape(lead)


@_on_device
def decode_sum(idx: torch.Tensor, embed: torch.Tensor, out_dtype=torch.float32, out=None, stages=None, accumulate=False) -> torch.Tensor:
    """idx [..., Q] int64, embed [Q, C, D] or [C, D] (shared by all Q) -> [..., D] = sum_q embed_q[idx_q].
    out (optional): a [..., D] tensor to write -- contiguous (one slice of a stacked [Q, ..., D] result) or a feature chunk of a wider
    contiguous tensor (rows at a uniform stride).
    stages = (q0, q1): only the stages q0 <= q < q1; accumulate: the sum continues from what `out` holds (fp32) -- decode_sum(stages=(0, Q - 1))
    followed by decode_sum(stages=(Q - 1, Q), accumulate=True) adds in the same order as one call (vqhip_decode_sum_range)."""
    _need_gpu(idx, embed)
    assert idx.dtype == torch.int64 and embed.dtype == torch.float32 and embed.is_contiguous()
    idx = idx.contiguous()
    Qall = idx.shape[-1]
    q0, q1 = (0, Qall) if stages is None else stages
    assert 0 <= q0 < q1 <= Qall
    Q = q1 - q0
    if embed.ndim == 2:
        C, D = embed.shape
        qstride = 0
    else:
        assert embed.shape[0] == Qall
        _, C, D = embed.shape
        qstride = C * D
    N = idx.numel() // Qall
    ldo = D
    if out is None:
        out = torch.empty(*idx.shape[:-1], D, dtype=out_dtype, device=idx.device)
    else:
        # contiguous, or a feature chunk of a wider contiguous tensor (a group's slice of GroupedResidualVQ's output: no torch.cat)
        out_dtype = out.dtype
        ok, oN, oD, ldo = as_rows(out)
        assert ok.data_ptr() == out.data_ptr() and oN == N and oD == D, "decode_sum: out must be row-addressable in place"
        assert out_dtype in (torch.float32, torch.bfloat16) and out.device == idx.device
    if N > 0:
        if stages is None and not accumulate:
            _check(lib().vqhip_decode_sum(_ptr(idx), N, Q, _ptr(embed), qstride, C, D, _ptr(out),
                                          F32 if out_dtype == torch.float32 else BF16, ldo, _stream()), "vqhip_decode_sum")
        else:
            assert not accumulate or out_dtype == torch.float32
            _check(lib().vqhip_decode_sum_range(ctypes.c_void_p(idx.data_ptr() + 8 * q0), Qall, N, Q, ctypes.c_void_p(embed.data_ptr() + 4 * q0 * qstride),
                                                qstride, C, D, _ptr(out), F32 if out_dtype == torch.float32 else BF16, ldo, int(bool(accumulate)),
                                                _stream()), "vqhip_decode_sum_range")
    return out


@_on_device
def score_indices(x: torch.Tensor, packed: torch.Tensor, embed2d: torch.Tensor, idx: torch.Tensor, *, cosine=False) -> torch.Tensor:
    """The reference-arithmetic score of code idx[n] for row n: cdist (Euclidean) or similarity (cosine; rows must already be
    unit-norm) -- what assign(want_best=True) reports for the winner, for searches that ran screened."""
    _need_gpu(x, packed, embed2d, idx)
    xk, N, D, ldx = as_rows(x)
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == N
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    if N > 0:
        _check(lib().vqhip_score_indices(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(packed), _ptr(embed2d), embed2d.shape[0],
                                         COSINE_PRENORM if cosine else EUCLID, _ptr(idx), _ptr(out), _stream()), "vqhip_score_indices")
    return out


@_on_device
def pack_best(best: torch.Tensor, local_index: torch.Tensor, index_offset: int, *, negate: bool) -> torch.Tensor:
    """(shard winner's score, index inside the shard) -> order-preserving int64 key for the MAX all-reduce (vqhip_pack_best)."""
    _need_gpu(best, local_index)
    assert best.dtype == torch.float32 and local_index.dtype == torch.int64 and best.numel() == local_index.numel()
    best, local_index = best.contiguous(), local_index.contiguous()
    key = torch.empty(best.shape, dtype=torch.int64, device=best.device)
    _check(lib().vqhip_pack_best(_ptr(best), _ptr(local_index), best.numel(), int(index_offset), int(negate), _ptr(key), _stream()), "vqhip_pack_best")
    return key


@_on_device
def unpack_best(key: torch.Tensor, lo: int, hi: int, *, negate: bool, want_best=False):
    """reduced keys -> (global index, index inside [lo, hi) or -1[, winning score]) in one launch (vqhip_unpack_best)."""
    _need_gpu(key)
    assert key.dtype == torch.int64
    key = key.contiguous()
    gidx, local = torch.empty_like(key), torch.empty_like(key)
    best = torch.empty(key.shape, dtype=torch.float32, device=key.device) if want_best else None
    _check(lib().vqhip_unpack_best(_ptr(key), key.numel(), int(lo), int(hi), int(negate), _ptr(gidx), _ptr(local), _ptr(best), _stream()), "vqhip_unpack_best")
    return (gidx, local, best) if want_best else (gidx, local)


TOPK_MAX = 8


def topk_supported(x: torch.Tensor, k: int, C: int) -> bool:
    xk, N, D, ldx = as_rows(x)
    return (1 <= k <= min(TOPK_MAX, C) and D in (32, 64, 128, 256, 512) and xk.dtype in (torch.float32, torch.bfloat16)
            and xk.data_ptr() % 16 == 0 and (ldx * xk.element_size()) % 16 == 0)


@_on_device
def topk(x: torch.Tensor, packed: torch.Tensor, C: int, k: int, *, cosine=False, skip_l2norm=False, want_values=False):
    """x [..., D] -> indices [..., k] (and values) of the k best codes in the reference's arithmetic, without the N x C tensor."""
    _need_gpu(x, packed)
    xk, N, D, ldx = as_rows(x)
    idx = torch.empty(*x.shape[:-1], k, dtype=torch.int64, device=x.device)
    val = torch.empty(*x.shape[:-1], k, dtype=torch.float32, device=x.device) if want_values else None
    if N > 0:
        metric = (COSINE_PRENORM if skip_l2norm else COSINE) if cosine else EUCLID
        _check(lib().vqhip_topk(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(packed), C, metric, k, _ptr(idx), _ptr(val), _stream()), "vqhip_topk")
    return (idx, val) if want_values else idx


@_on_device
def expire_scatter(cluster_size: torch.Tensor, embed_avg: torch.Tensor, embed: torch.Tensor, candidates: torch.Tensor, threshold: float,
                   reset: float, n_expired: torch.Tensor | None = None):
    """In place on one codebook's (cluster_size [C], embed_avg [C, D], embed [C, D]): the j-th expired code takes candidates[j]."""
    _need_gpu(cluster_size, embed_avg, embed, candidates)
    C, D = embed.shape
    assert candidates.shape == (C, D) and candidates.dtype == torch.float32 and candidates.is_contiguous()
    for t in (cluster_size, embed_avg, embed):
        assert t.is_contiguous() and t.dtype == torch.float32
    _check(lib().vqhip_expire_scatter(_ptr(cluster_size), _ptr(embed_avg), _ptr(embed), _ptr(candidates), C, D, float(threshold),
                                      float(reset), _ptr(n_expired), _stream()), "vqhip_expire_scatter")


_PRIMES = {}


def _next_prime(n: int) -> int:
    """smallest prime >= n (cached per batch size)"""
    p = _PRIMES.get(n)
    if p is None:
        p = max(n, 2)
        while any(p % d == 0 for d in range(2, int(p ** 0.5) + 1)):
            p += 1
        _PRIMES[n] = p
    return p


@_on_device
def expire_pick(cluster_size: torch.Tensor, embed_avg: torch.Tensor, embed: torch.Tensor, rows: torch.Tensor, threshold: float, reset: float,
                *, cosine=False):
    """In place on one codebook: every code with cluster_size < threshold takes a row of `rows` [n, D] (float32 / bfloat16), distinct
    rows for distinct codes when C <= n; two draws of torch's generator on the device, one kernel, no host read (vqhip_expire_pick)."""
    _need_gpu(cluster_size, embed_avg, embed, rows)
    C, D = embed.shape
    rk, n, Dr, ldx = as_rows(rows)
    assert Dr == D and n >= 1
    for t in (cluster_size, embed_avg, embed):
        assert t.is_contiguous() and t.dtype == torch.float32
    p = _next_prime(n)
    ab = torch.randint(1, p, (2,), device=rows.device, dtype=torch.int64)
    _check(lib().vqhip_expire_pick(_ptr(cluster_size), _ptr(embed_avg), _ptr(embed), _ptr(rk), _dtype_code(rk), n, ldx, _ptr(ab), p, C, D,
                                   float(threshold), float(reset), int(cosine), _stream()), "vqhip_expire_pick")


@_on_device
def kmeans_update(means: torch.Tensor, embed_sum: torch.Tensor, count: torch.Tensor, *, cosine=False):
    """In place: means[c] = embed_sum[c] / count[c] for non-empty bins (l2-normalised if cosine)."""
    _need_gpu(means, embed_sum, count)
    C, D = means.shape
    for t in (means, embed_sum, count):
        assert t.is_contiguous() and t.dtype == torch.float32
    _check(lib().vqhip_kmeans_update(_ptr(means), _ptr(embed_sum), _ptr(count), C, D, int(cosine), _stream()), "vqhip_kmeans_update")


@_on_device
def row_sumsq(x: torch.Tensor) -> torch.Tensor:
    _need_gpu(x)
    xk, N, D, ldx = as_rows(x)
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    if N > 0:
        _check(lib().vqhip_row_sumsq(_ptr(xk), _dtype_code(xk), N, D, ldx, _ptr(out), _stream()), "vqhip_row_sumsq")
    return out
