"""Multi-GPU pieces of the hot path (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on MI355X, "gloo" in the CPU tests).

1. Data parallel (what the reference does, vqp.py:603/607): handled inside `Codebook.quantize` with ONE
   all-reduce of the fused `embed_sum || count` buffer.
2. Codebook-sharded argmin (new capability, BASELINE config 4): rank p owns codes
   [p*C/P, (p+1)*C/P).  Every rank scores the same rows against its shard with `vqhip_assign`, packs
   (score, global index) into an order-preserving int64 key and ONE `all_reduce(MAX)` of N x 8 bytes
   yields the global winner with the reference's tie rule (lowest index among equal scores).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

_LOW = 0xFFFFFFFF


def pack_score_index(score: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """(fp32 score to MAXIMISE, int64 global index < 2^32) -> int64 key; key order == (score, -index) order.

    High word: the IEEE-754 bits mapped to a signed int32 that sorts like the float (negative floats have
    their magnitude bits flipped).  Low word: 0xFFFFFFFF - index, so that among equal scores the LOWEST
    index has the LARGEST key -- ATen argmax's first-occurrence rule (vqp.py:140).  torch has no uint64
    collectives, hence the signed construction."""
    bits = score.contiguous().view(torch.int32)
    ordered = torch.where(bits < 0, bits ^ 0x7FFFFFFF, bits).to(torch.int64)
    return (ordered << 32) | (_LOW - index.to(torch.int64))


def unpack_score_index(key: torch.Tensor):
    index = _LOW - (key & _LOW)
    ordered = (key >> 32).to(torch.int32)
    bits = torch.where(ordered < 0, ordered ^ 0x7FFFFFFF, ordered)
    return bits.view(torch.float32), index


def merge_sharded_argmin(best: torch.Tensor, local_index: torch.Tensor, shard_offset: int, *, euclid: bool,
                         group=None):
    """best: winning distance (euclid) or similarity (cosine) of THIS rank's shard, local_index its index
    inside the shard.  Returns (global index int64, global best fp32), identical on every rank."""
    score = -best if euclid else best
    key = pack_score_index(score, local_index + shard_offset)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(key, op=dist.ReduceOp.MAX, group=group)
    s, idx = unpack_score_index(key)
    return idx, (-s if euclid else s)


def shard_bounds(C: int, world: int, rank: int):
    per = (C + world - 1) // world
    lo = min(rank * per, C)
    return lo, min(lo + per, C)


def fused_stats_allreduce(embed_sum: torch.Tensor, count: torch.Tensor, group=None):
    """embed_sum [C, D] and count [C] must be views of ONE contiguous buffer (Codebook.quantize allocates
    them that way); a single SUM all-reduce instead of the reference's two (vqp.py:603, 607)."""
    base = embed_sum.untyped_storage().data_ptr()
    assert count.untyped_storage().data_ptr() == base, "embed_sum and count must share one buffer"
    flat = torch.as_strided(embed_sum, (embed_sum.numel() + count.numel(),), (1,), embed_sum.storage_offset())
    dist.all_reduce(flat, group=group)
