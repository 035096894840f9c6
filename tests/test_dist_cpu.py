"""CPU, world_size = 2, gloo: the N > 1 logic of the path -- fused statistics all-reduce (data parallel) and
the codebook-sharded argmin merge -- against a single-process run on the concatenated batch / full codebook.
Device compute is replaced by the oracle here (these tests check the collective logic, not the kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vq_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _dp_worker(rank, world, port, out):
    from vector_quantize_pytorch_amd.parallel import fused_stats_allreduce
    _init(rank, world, port)
    g = torch.Generator().manual_seed(0)
    N, C, D = 1024, 64, 32
    x = torch.randn(N, D, generator=g); e = torch.randn(C, D, generator=g)
    xs = x[rank * N // world:(rank + 1) * N // world]
    idx, _ = O.c_assign(xs, e)
    cnt, es = O.c_ema_stats(xs, idx, C)
    buf = torch.zeros(C * D + C)
    buf[:C * D] = es.reshape(-1); buf[C * D:] = cnt
    esum, count = buf[:C * D].view(C, D), buf[C * D:]
    fused_stats_allreduce(esum, count)
    if rank == 0:
        torch.save(dict(esum=esum.clone(), count=count.clone()), out)
    dist.destroy_process_group()


def test_dp_fused_stats_allreduce_equals_single_process(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    r = torch.load(out)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1024, 32, generator=g); e = torch.randn(64, 32, generator=g)
    idx, _ = O.c_assign(x, e)
    cnt, es = O.c_ema_stats(x, idx, 64)
    assert torch.equal(r["count"], cnt)
    assert (r["esum"] - es).abs().max().item() <= 1e-5 * es.abs().max().item()


def _shard_worker(rank, world, port, out, cosine):
    from vector_quantize_pytorch_amd.parallel import merge_sharded_argmin, shard_bounds
    _init(rank, world, port)
    g = torch.Generator().manual_seed(1)
    N, C, D = 2000, 100, 48            # C not divisible by the world size on purpose (100 -> 50/50; 3 ranks would be ragged)
    x = torch.randn(N, D, generator=g); e = torch.randn(C, D, generator=g)
    e[70] = e[10]; e[99] = e[10]       # duplicated codes across shards: the lowest global index must win
    if cosine:
        x, e = O.l2norm(x), O.l2norm(e)
    lo, hi = shard_bounds(C, world, rank)
    idx_l, best_l = O.c_assign(x, e[lo:hi].contiguous(), cosine)
    idx, best = merge_sharded_argmin(best_l, idx_l, lo, euclid=not cosine)
    if rank == 1:
        torch.save(dict(idx=idx, best=best), out)
    dist.destroy_process_group()


@pytest.mark.parametrize("cosine", [False, True])
def test_sharded_argmin_merge_equals_full_codebook(tmp_path, cosine):
    world, port, out = 2, _free_port(), str(tmp_path / "sh.pt")
    mp.spawn(_shard_worker, args=(world, port, out, cosine), nprocs=world, join=True)
    r = torch.load(out)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2000, 48, generator=g); e = torch.randn(100, 48, generator=g)
    e[70] = e[10]; e[99] = e[10]
    if cosine:
        x, e = O.l2norm(x), O.l2norm(e)
    idx, best = O.c_assign(x, e, cosine)
    assert torch.equal(r["idx"], idx)
    assert torch.equal(r["best"], best)
    assert not ((idx == 70) | (idx == 99)).any()


def _gather_worker(rank, world, port, out, chunks):
    from vector_quantize_pytorch_amd.parallel import gathered_search, shard_bounds
    _init(rank, world, port)
    g = torch.Generator().manual_seed(5)
    n, C, D = 1003, 100, 32            # 1003 rows per rank: chunks of unequal size, the last one shorter
    x = torch.randn(world, n, D, generator=g); e = torch.randn(C, D, generator=g)
    e[70] = e[10]                      # a duplicate across the shards: the lowest global index must win in every chunking
    lo, hi = shard_bounds(C, world, rank)
    shard = e[lo:hi].contiguous()

    def search(rows):
        idx, best = O.c_assign(rows, shard)
        return best, idx

    parts, gidx, local = gathered_search(x[rank].contiguous(), search, (lo, hi), euclid=True, chunks=chunks)
    rows = torch.cat([p.reshape(world, -1, D) for p, _ in parts], dim=1)       # the gathered chunks back in row order
    assert torch.equal(rows, x) and [o for _, o in parts] == sorted(o for _, o in parts)
    if rank == 0:
        torch.save(dict(gidx=gidx, local=local), out)
    dist.destroy_process_group()


def test_chunked_row_gather_of_the_sharded_search_returns_the_unchunked_indices(tmp_path):
    """parallel.gathered_search (round 6): the rows of all ranks all-gathered in K chunks, chunk k + 1 travelling and chunk k - 1's
    keys being MAX-reduced while chunk k is searched -- K = 1, 2, 4 (ragged chunks) give the same global and local indices as the
    search of the whole batch against the full codebook."""
    world = 2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(world, 1003, 32, generator=g); e = torch.randn(100, 32, generator=g)
    e[70] = e[10]
    want, _ = O.c_assign(x.reshape(-1, 32), e)
    for K in (1, 2, 4):
        port, out = _free_port(), str(tmp_path / f"g{K}.pt")
        mp.spawn(_gather_worker, args=(world, port, out, K), nprocs=world, join=True)
        r = torch.load(out)
        assert torch.equal(r["gidx"], want), f"chunks={K}"
        lo, hi = 0, 50
        assert torch.equal(r["local"], torch.where((want >= lo) & (want < hi), want - lo, torch.full_like(want, -1)))
    assert not (want == 70).any()


def _sample_worker(rank, world, port, out):
    from vector_quantize_pytorch_amd.codebook import sample_rows_distributed
    _init(rank, world, port)
    torch.manual_seed(rank)
    local = torch.full((1, 10 + 20 * rank, 4), float(rank))
    s = sample_rows_distributed(local, 16)
    assert s.shape == (1, 16, 4)
    gathered = [torch.empty_like(s) for _ in range(world)]
    dist.all_gather(gathered, s)
    assert all(torch.equal(gathered[0], t) for t in gathered)      # every rank ends with the same sample
    if rank == 0:
        torch.save(s, out)
    dist.destroy_process_group()


def test_distributed_sampling_is_consistent_across_ranks(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "s.pt")
    mp.spawn(_sample_worker, args=(world, port, out), nprocs=world, join=True)
    s = torch.load(out)
    assert set(s.unique().tolist()) <= {0.0, 1.0}
