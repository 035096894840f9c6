"""MI355X codebook: the object `VectorQuantize._codebook` holds.

Mirrors the interface of the reference's `Codebook` (vector_quantize_pytorch.py:349-791): same
constructor keywords, same buffers / state_dict keys (`initted`, `cluster_size`, `embed_avg`,
`embed`, :415-423), same public methods other code touches (`transform_input`, `update_ema`,
`expire_codes_`, `update_indices`, `forward`).  Everything beneath is libvqhip.so (csrc/vqhip.hip):
the N x C distance / one-hot tensors the reference materialises never exist here.
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import Tensor, nn

from . import _lib as L


def _l2norm(t, eps=1e-6):
    return F.normalize(t, p=2, dim=-1, eps=eps)


def _kaiming_uniform(*shape):
    t = torch.empty(shape)
    nn.init.kaiming_uniform_(t)
    return t


def sample_rows(samples: Tensor, num: int) -> Tensor:
    """`num` rows of `samples` [n, d]; RNG consumption identical to the reference's sample_vectors
    (vqp.py:156-163) so that a shared seed gives the same draw on the same device."""
    n = samples.shape[0]
    if n >= num:
        pick = torch.randperm(n, device=samples.device)[:num]
    else:
        pick = torch.randint(0, n, (num,), device=samples.device)
    return samples[pick]


def batched_sample_rows(samples: Tensor, num: int) -> Tensor:
    return torch.stack([sample_rows(s, num) for s in samples.unbind(0)], 0)


def _all_gather_sizes(n: int, device) -> list:
    size = torch.tensor(n, dtype=torch.long, device=device)
    sizes = [torch.empty_like(size) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, size)
    return [int(s) for s in sizes]


def sample_rows_distributed(samples: Tensor, num: int) -> Tensor:
    """Cross-rank sampling for k-means seeding / dead-code replacement (vqp.py:211-229 semantics:
    a multinomial split of `num` over ranks proportional to their row counts, local draws, then an
    all-gather).  samples [1, n, d] -> [1, num, d]."""
    local = samples[0]
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = _all_gather_sizes(local.shape[0], local.device)
    split = torch.empty(world, dtype=torch.long, device=local.device)
    if rank == 0:
        probs = torch.tensor(sizes, dtype=torch.float32)
        probs = probs / probs.sum()
        left, rem = num, 1.0
        for i in range(world):
            if i == world - 1:
                k = left
            else:
                k = int(torch.binomial(torch.tensor(float(left)), (probs[i] / rem).clamp(0, 1)).item())
            split[i] = k
            left -= k
            rem -= float(probs[i])
    dist.broadcast(split, src=0)
    counts = split.tolist()
    mine = sample_rows(local, counts[rank])
    # one padded all-gather instead of `world` variably-sized broadcasts
    width = max(max(counts), 1)
    pad = torch.zeros(width, local.shape[-1], dtype=local.dtype, device=local.device)
    pad[: mine.shape[0]] = mine
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad)
    out = torch.cat([g[:c] for g, c in zip(gathered, counts)], 0)
    return out[None]


def fused_stats_allreduce(embed_sum: torch.Tensor, count: torch.Tensor, group=None):
    """embed_sum [C, D] and count [C] must be views of ONE contiguous buffer (Codebook.quantize allocates
    them that way); a single SUM all-reduce instead of the reference's two (vqp.py:603, 607)."""
    base = embed_sum.untyped_storage().data_ptr()
    assert count.untyped_storage().data_ptr() == base, "embed_sum and count must share one buffer"
    flat = torch.as_strided(embed_sum, (embed_sum.numel() + count.numel(),), (1,), embed_sum.storage_offset())
    dist.all_reduce(flat, group=group)


class Codebook(nn.Module):
    def __init__(
        self,
        dim,
        codebook_size,
        num_codebooks=1,
        kmeans_init=False,
        kmeans_iters=10,
        sync_kmeans=True,
        decay=0.8,
        eps=1e-5,
        threshold_ema_dead_code=2,
        reset_cluster_size=None,
        use_ddp=False,
        learnable_codebook=False,
        gumbel_sample=None,
        sample_codebook_temp=1.,
        ema_update=True,
        manual_ema_update=False,
        affine_param=False,
        sync_affine_param=False,
        affine_param_batch_decay=0.99,
        affine_param_codebook_decay=0.9,
        use_cosine_sim=False,
        vq_bridge: Optional[nn.Module] = None,
    ):
        super().__init__()
        if not (1 <= dim <= L.WIDE_MAX_DIM):
            raise NotImplementedError(f"codebook dim {dim}: the HIP path supports 1 <= dim <= {L.WIDE_MAX_DIM} (the tuned kernels up to 512, "
                                      f"plain exact ones beyond: csrc/vq_wide.hip)")

        self.dim = dim
        self.codebook_size = codebook_size
        self.num_codebooks = num_codebooks
        self.use_cosine_sim = use_cosine_sim
        self.transform_input = _l2norm if use_cosine_sim else (lambda t: t)

        self.decay = decay
        self.eps = eps
        self.ema_update = ema_update
        self.manual_ema_update = manual_ema_update
        self.kmeans_iters = kmeans_iters
        self.threshold_ema_dead_code = threshold_ema_dead_code
        self.has_dead_code_replacement = threshold_ema_dead_code > 0
        self.reset_cluster_size = threshold_ema_dead_code if reset_cluster_size is None else reset_cluster_size
        self.sample_codebook_temp = sample_codebook_temp
        self.learnable_codebook = learnable_codebook
        self.affine_param = affine_param
        self.sync_affine_param = sync_affine_param
        self.affine_param_batch_decay = affine_param_batch_decay
        self.affine_param_codebook_decay = affine_param_codebook_decay
        self.vq_bridge = vq_bridge

        self.use_ddp = use_ddp
        assert not (use_ddp and num_codebooks > 1 and kmeans_init), \
            'kmeans init is not compatible with multiple codebooks in distributed environment for now'
        sync_sampling = use_ddp and sync_kmeans
        self.sample_fn = sample_rows_distributed if sync_sampling else batched_sample_rows
        self.replace_sample_fn = sample_rows_distributed if sync_sampling else batched_sample_rows
        self.sync_kmeans_stats = sync_sampling

        if kmeans_init:
            embed = torch.zeros(num_codebooks, codebook_size, dim)
        else:
            embed = _kaiming_uniform(num_codebooks, codebook_size, dim)     # vqp.py:385
            if use_cosine_sim:
                embed = _l2norm(embed)

        self.register_buffer('initted', torch.tensor(not kmeans_init))
        self.register_buffer('cluster_size', torch.ones(num_codebooks, codebook_size))
        self.register_buffer('embed_avg', embed.clone())
        if learnable_codebook:                     # vqp.py:419-423: same state_dict key, Parameter instead of buffer
            self.embed = nn.Parameter(embed)
        else:
            self.register_buffer('embed', embed)

        self._initted_known = not kmeans_init     # python-side cache: no host sync per forward
        self._affine_needs_init = {}              # same idea for the affine_param moment buffers

        if affine_param:                           # vqp.py:442-448: same buffer names / state_dict keys
            self.register_buffer('batch_mean', None)
            self.register_buffer('batch_variance', None)
            self.register_buffer('codebook_mean_needs_init', torch.tensor(True))
            self.register_buffer('codebook_mean', torch.empty(num_codebooks, 1, dim))
            self.register_buffer('codebook_variance_needs_init', torch.tensor(True))
            self.register_buffer('codebook_variance', torch.empty(num_codebooks, 1, dim))

    # ---- affine reparametrisation of the codebook (vqp.py:475-542): running first / second moments of the batch and
    #      of the codebook; the codebook is searched after being mapped onto the batch statistics ----
    def _decayed(self, name: str, new: Tensor, decay: float):
        """update_with_decay (vqp.py:475-492) without a host sync per step: whether the statistic still needs its first
        value is cached on the Python side (re-read from the `*_needs_init` buffer once after a checkpoint load), and the
        running value is updated in place, so external references to the buffer stay valid."""
        old = getattr(self, name)
        flag = getattr(self, name + '_needs_init', None)
        needs_init = self._affine_needs_init.get(name)
        if needs_init is None:
            needs_init = old is None or (flag is not None and bool(flag))
        if needs_init:
            if old is None:
                self.register_buffer(name, new.detach().clone())
            else:
                old.copy_(new.detach())
            if flag is not None:
                flag.fill_(False)
            self._affine_needs_init[name] = False
            return
        self._affine_needs_init[name] = False
        old.mul_(decay).add_(new.detach() * (1 - decay))      # == old * decay + new * (1 - decay), same roundings

    @torch.no_grad()
    def update_affine(self, data: Tensor, embed: Tensor, mask: Optional[Tensor] = None):
        if self.training:
            self._decayed('codebook_mean', embed.mean(dim=1, keepdim=True), self.affine_param_codebook_decay)
            self._decayed('codebook_variance', embed.var(dim=1, unbiased=False, keepdim=True), self.affine_param_codebook_decay)
        if mask is not None:
            data = data[mask].reshape(data.shape[0], -1, data.shape[-1])
        if not self.sync_affine_param:
            self._decayed('batch_mean', data.mean(dim=1, keepdim=True), self.affine_param_batch_decay)
            self._decayed('batch_variance', data.var(dim=1, unbiased=False, keepdim=True), self.affine_param_batch_decay)
            return
        n = torch.tensor(float(data.shape[-2]), device=data.device, dtype=data.dtype)
        dist.all_reduce(n)
        total = data.sum(dim=1, keepdim=True)
        dist.all_reduce(total)
        mean = total / n
        self._decayed('batch_mean', mean, self.affine_param_batch_decay)
        sq = ((data - mean) ** 2).sum(dim=1, keepdim=True)
        dist.all_reduce(sq)
        self._decayed('batch_variance', sq / n, self.affine_param_batch_decay)

    # ---- state handling --------------------------------------------------------------------------
    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._initted_known = None                # re-read `initted` lazily after a checkpoint load
        self._affine_needs_init = {}

    def _is_initted(self) -> bool:
        if self._initted_known is None:
            self._initted_known = bool(self.initted.item())
        return self._initted_known

    def _views(self, h: int):
        return self.cluster_size[h], self.embed_avg[h], self.embed.data[h]

    # ---- k-means initialisation (vqp.py:238-278, 450-473) -----------------------------------------
    @torch.no_grad()
    def init_embed_(self, flat: Tensor, mask: Optional[Tensor] = None):
        """flat [H, n, d] (already l2-normalised when cosine)."""
        if self._is_initted():
            return
        H, C, D = self.num_codebooks, self.codebook_size, self.dim
        data = flat.float()
        if mask is not None:
            data = data[mask].reshape(H, -1, D)
        means = self.sample_fn(data, C).contiguous()
        bins = None
        for _ in range(self.kmeans_iters):
            new_means, bins = [], []
            for h in range(H):
                m = means[h].contiguous()
                # means may be zero rows when cosine; the kernel's metric handles both
                r = L.assign(data[h], L.pack_codebook(m), m, cosine=self.use_cosine_sim, want_q=False,
                             skip_l2norm=True)
                cnt, esum = L.ema_accumulate(data[h], r["idx"], C)
                if not self.sync_kmeans_stats and not self.use_cosine_sim:
                    L.kmeans_update(m, esum, cnt)             # m <- esum / cnt where cnt > 0 (vqhip_kmeans_update), in place
                    new_means.append(m)
                    bins.append(cnt)
                    continue
                if self.sync_kmeans_stats:
                    dist.all_reduce(cnt)
                    # the reference divides by the all-reduced bins *before* all-reducing the means
                    # (vqp.py:258-267); keep that order
                zero = cnt == 0
                nm = esum / cnt.masked_fill(zero, 1.)[:, None]
                if self.sync_kmeans_stats:
                    dist.all_reduce(nm)
                if self.use_cosine_sim:
                    nm = _l2norm(nm)
                new_means.append(torch.where(zero[:, None], m, nm))
                bins.append(cnt)
            means = torch.stack(new_means)
            bins = torch.stack(bins)
        self.embed_avg.copy_(means * bins[..., None])
        self.cluster_size.copy_(bins)
        self.update_ema()
        self.initted.fill_(True)
        self._initted_known = True

    # ---- EMA pieces -------------------------------------------------------------------------------
    @torch.no_grad()
    def update_ema(self):
        """embed <- embed_avg / laplace-smoothed cluster size (vqp.py:576-584)."""
        for h in range(self.num_codebooks):
            cs, ea, e = self._views(h)
            L.ema_finalize(cs, ea, e, None, None, decay=self.decay, eps=self.eps, cosine=self.use_cosine_sim,
                           do_lerp=False, do_update_ema=True)

    @torch.no_grad()
    def replace(self, batch_samples: Tensor, batch_mask: Tensor, seq_mask: Optional[Tensor] = None):
        """vqp.py:544-562: overwrite expired codes with sampled batch rows."""
        if self.use_cosine_sim:
            batch_samples = _l2norm(batch_samples)
        for h in range(batch_samples.shape[0]):
            samples, m = batch_samples[h], batch_mask[h]
            if seq_mask is not None:
                samples = samples[seq_mask[h]]
            if samples.numel() == 0:
                continue
            picked = self.replace_sample_fn(samples[None], int(m.sum().item()))[0].to(self.embed.dtype)
            self.embed.data[h][m] = picked
            self.cluster_size[h][m] = self.reset_cluster_size
            self.embed_avg[h][m] = picked * self.reset_cluster_size

    expire_without_host_sync = False     # class default; set True (or capture the step in a HIP graph) for the device-side path

    @torch.no_grad()
    def expire_codes_(self, batch_samples: Tensor, seq_mask: Optional[Tensor] = None, any_expired: Optional[bool] = None):
        """any_expired: the caller has read `any(cluster_size < threshold)` already (ResidualVQ reads it for all its layers with ONE
        host sync, Codebook.any_expired_many) -- skips this call's own read.
        vqp.py:564-574.  Default path: the reference's control flow -- `any(expired)` and the number of expired codes are read
        on the host (two syncs, only when threshold_ema_dead_code > 0; 0 is VectorQuantize's default, vqp.py:818), which keeps
        torch's generator in lock-step with the reference (no draw when nothing expired).  Device-side path
        (`expire_without_host_sync`, or automatically while the stream is being captured into a graph): every step draws one random
        permutation of the batch rows (two generator draws) and vqhip_expire_pick hands every expired code its row -- rows without
        replacement as in the reference, one launch, no host round trip, but the generator advances on steps without expired codes
        too."""
        if not self.has_dead_code_replacement or not self.training:
            return
        H = batch_samples.shape[0]
        samples = batch_samples.reshape(H, -1, batch_samples.shape[-1])
        nosync = (self.expire_without_host_sync or os.environ.get("VQHIP_EXPIRE_DEVICE", "0") == "1"
                  or torch.cuda.is_current_stream_capturing()) and seq_mask is None and self.replace_sample_fn is batched_sample_rows
        if nosync:
            # ONE launch per codebook (vqhip_expire_pick): expired code c takes row pi(c) of the batch, pi an affine permutation drawn
            # from torch's generator -- rows without replacement like sample_vectors (vqp.py:180-188), nothing sized by the count
            for h in range(H):
                if samples[h].shape[0] == 0:
                    continue
                cs, ea, e = self._views(h)
                L.expire_pick(cs, ea, e, samples[h], self.threshold_ema_dead_code, self.reset_cluster_size, cosine=self.use_cosine_sim)
            return
        if any_expired is False:
            return
        expired = self.cluster_size < self.threshold_ema_dead_code
        if any_expired is None and not bool(expired.any()):
            return
        self.replace(samples.float(), expired, seq_mask)

    def expiry_reads_host(self) -> bool:
        """does expire_codes_ take the reference's control flow (a host read of any(expired)) in this state?"""
        return bool(self.has_dead_code_replacement and self.training
                    and not (self.expire_without_host_sync or os.environ.get("VQHIP_EXPIRE_DEVICE", "0") == "1"
                             or torch.cuda.is_current_stream_capturing()))

    @staticmethod
    def any_expired_many(codebooks) -> list:
        """[any(cluster_size < threshold) for each codebook] with one device-to-host copy for all of them (the reference reads it per
        layer, vqp.py:570: eight pipeline drains per step of an 8-stage residual VQ)"""
        flags = torch.stack([(cb.cluster_size < cb.threshold_ema_dead_code).any() for cb in codebooks])
        return flags.tolist()

    def _fold_stats(self, h, count, esum, ema_update_weight, accum_ema_update, ema_update):
        cs, ea, e = self._views(h)
        w = ema_update_weight
        if callable(w):
            w = w(esum[None], count[None])
        if torch.is_tensor(w):
            w = w.reshape(-1, self.codebook_size)[h if w.numel() > self.codebook_size else 0]
            w = w.to(torch.float32).contiguous()
        elif w is not None:
            raise NotImplementedError("ema_update_weight must be a tensor [codebook_size], a callable or None")
        if accum_ema_update:                       # vqp.py:612-614: park the statistics for a later fold
            for buf, new in ((self.cluster_size, count), (self.embed_avg, esum)):
                if buf.grad is None:
                    buf.grad = torch.zeros_like(buf)
                buf.grad[h] += new
            return
        if self.cluster_size.grad is not None:      # vqp.py:80-82: fold parked statistics first
            count = count + self.cluster_size.grad[h]
            esum = esum + self.embed_avg.grad[h]
            if h == self.num_codebooks - 1:         # ... "and set it to None": folded once, not on every later step
                self.cluster_size.grad = None
                self.embed_avg.grad = None
        L.ema_finalize(cs, ea, e, count, esum, decay=self.decay, eps=self.eps, cosine=self.use_cosine_sim, weight=w,
                       do_lerp=True, do_update_ema=bool(ema_update and not self.manual_ema_update))

    # ---- the hot path -----------------------------------------------------------------------------
    @torch.no_grad()
    def quantize(self, x: Tensor, *, mask: Optional[Tensor] = None, freeze_codebook=False,
                 ema_update_weight=None, accum_ema_update=False, ema_update=None, update_usage=True,
                 want_sqerr=False, input_normalized=False, q_out=None, embed_override=None, want_q=True, loss_scale=None):
        """x [b, n, d] (or [h, b, n, d] when num_codebooks > 1), float32 / bfloat16, RAW input: for the
        cosine metric the l2norm of vqp.py:1159 is fused into the kernel (pass input_normalized=True
        when x is already unit-norm).  Returns dict(q, idx, sqerr_partials, nblk, rnorm); with loss_scale given the fused
        train step may serve the call, and the dict then carries `loss` (= loss_scale * sum of squared errors) instead of partials."""
        ema_update = self.ema_update if ema_update is None else ema_update
        H, C = self.num_codebooks, self.codebook_size
        xs = x if x.ndim == 4 else x[None]
        assert xs.shape[0] == H and xs.shape[-1] == self.dim
        rmask = None if mask is None else mask.reshape(-1)      # same rows for every codebook

        if not self._is_initted():
            flat = xs.reshape(H, -1, self.dim).float()
            if self.use_cosine_sim and not input_normalized:
                flat = _l2norm(flat)
            self.init_embed_(flat, None if rmask is None else rmask[None].expand(H, -1))

        do_update = (self.training and update_usage and not freeze_codebook
                     and (ema_update or self.has_dead_code_replacement))
        x_stats = xs
        if self.affine_param:                                   # vqp.py:705-706, 721-724, 594-597
            flat = xs.reshape(H, -1, self.dim).float()
            self.update_affine(flat, self.embed, None if rmask is None else rmask[None].expand(H, -1).bool())
            cstd = self.codebook_variance.clamp(min=1e-5).sqrt()
            bstd = self.batch_variance.clamp(min=1e-5).sqrt()
            base = self.embed if embed_override is None else embed_override
            embed_override = (base.detach() - self.codebook_mean) * (bstd / cstd) + self.batch_mean
            if do_update:
                x_stats = ((flat - self.batch_mean) * (cstd / bstd) + self.codebook_mean).reshape(xs.shape)
        # the whole training forward of the plain case as ONE library call (vqhip_vq_train_step): pack, search, statistics, the
        # commitment loss' squared error and -- without a collective in between -- the EMA fold
        if (H == 1 and do_update and want_sqerr and loss_scale is not None and ema_update
                and not self.affine_param and embed_override is None and ema_update_weight is None and not accum_ema_update
                and not self.manual_ema_update and self.cluster_size.grad is None and self.embed.dtype == torch.float32
                and L.vq_step_supported(xs[0], C)):
            cs, ea, e = self._views(0)
            x0 = xs[0]
            if self.use_cosine_sim and not input_normalized:    # vqp.py:1157-1159 (the loss below then compares with the unit-norm rows,
                x0 = L.l2norm_rows(x0)                          #  as the reference's does)
            r = L.vq_train_step(x0, e, ea, cs, decay=self.decay, eps=self.eps, want_q=want_q, q_out=q_out, loss_scale=loss_scale,
                                fold=not self.use_ddp, cosine=self.use_cosine_sim, row_mask=rmask,
                                # (a persistent workspace per (stream, size) instead of a cached allocator block per forward: built for
                                #  VERDICT r4 #9, measured no gain -- 1 024-row step 132 vs 137 us, 16 384 rows 186 vs 190, 2^20 rows 877 vs
                                #  876, profiles/r5_final/scale_n_scratch_cache.txt: the allocator's cached block costs what the lookup costs --
                                #  so it stays opt-in)
                                reuse_scratch=os.environ.get("VQHIP_SCRATCH_CACHE", "0") == "1")
            if self.use_ddp:
                dist.all_reduce(r["stats"])   # ONE collective for count || embed_sum (RCCL over xGMI), then the fold
                self._fold_stats(0, r["count"], r["embed_sum"], None, False, ema_update)
            self.expire_codes_(xs.reshape(H, -1, self.dim), seq_mask=None if rmask is None else rmask[None].bool())
            return dict(q=r["q"], idx=r["idx"], sqerr_partials=None, nblk=0, rnorm=None, loss=r["loss"], n_exact=r["n_exact"], n_pair=r["n_pair"])
        outs = []
        # Several heads with their own codebooks (vqp.py:1044-1049; the reference runs them as one batched einsum over h): ONE set of
        # launches for the H packs and the H searches (vqhip_pack_codebook_batched / vqhip_assign_screened_batched, grid dimension y =
        # head) instead of H x (2 + 4).  Needs the screened search for every head and no squared error from the search kernel (the
        # statistics pass sums it when it runs on the searched rows); everything else keeps the per-head loop below.
        rb = packed_all = None
        xs_raw = xs                     # (dead-code replacement below samples the rows as they came in, like the per-head loop)
        E_all = (self.embed if embed_override is None else embed_override).detach()
        # (cosine: on unit-norm rows -- handed in, or normalised below -- the loss' squared error is the Euclidean one of those rows)
        stats_sums_loss = (want_sqerr and do_update and x_stats is xs and L.stats_sqerr_supported(xs[0])
                           and (not self.use_cosine_sim or input_normalized or L.screen_supported(xs[0], C)))
        if H > 1 and not self.affine_param and (not want_sqerr or stats_sums_loss) and E_all.dtype == torch.float32:
            xs_b = xs
            if self.use_cosine_sim and not input_normalized and L.screen_supported(xs[0], C):
                xs_b = L.l2norm_rows(xs)            # (rows are independent: every head's rows in one launch, vqp.py:37-38 at :1159)
            prenorm_b = input_normalized or xs_b is not xs
            if L.assign_batched_supported(xs_b, C, cosine=self.use_cosine_sim, skip_l2norm=prenorm_b):
                E_all = E_all.contiguous()
                packed_all = L.pack_codebook_batched(E_all)
                rb = L.assign_batched(xs_b, packed_all, E_all, cosine=self.use_cosine_sim, skip_l2norm=prenorm_b, want_q=want_q,
                                      want_rnorm=self.use_cosine_sim and not prenorm_b, row_mask=rmask)
                if xs_b is not xs:
                    xs = x_stats = xs_b
                    input_normalized = True
        if (rb is not None and do_update and rb["rnorm"] is None and x_stats is xs and embed_override is None and ema_update_weight is None
                and not accum_ema_update and self.cluster_size.grad is None and self.embed_avg.grad is None):
            # ... and the H statistics passes and folds as well (vqhip_ema_accumulate_batched / vqhip_ema_finalize_batched): one
            # [H, C D + C] buffer, ONE all-reduce for all heads under data parallelism (the reference: two per head)
            stats = torch.zeros(H, (C * self.dim + C + 3) // 4 * 4, dtype=torch.float32, device=x.device)
            parts = L.ema_accumulate_batched(xs, rb["idx"].reshape(H, -1), C, stats, row_mask=rmask,
                                             sqerr_from=(packed_all, E_all) if stats_sums_loss else None)
            if self.use_ddp:
                dist.all_reduce(stats)
            L.ema_finalize_batched(self.cluster_size, self.embed_avg, self.embed.data, stats, decay=self.decay, eps=self.eps,
                                   cosine=self.use_cosine_sim, do_update_ema=bool(ema_update and not self.manual_ema_update))
            self.expire_codes_(xs_raw.reshape(H, -1, self.dim), seq_mask=None if rmask is None else rmask[None].expand(H, -1).bool())
            return dict(q=rb["q"], idx=rb["idx"], sqerr_partials=None if parts is None else parts.reshape(-1),
                        nblk=0 if parts is None else parts.numel(), rnorm=None)
        for h in range(H):
            # embed_override: the codebook actually searched when it is a function of the stored one (vq_bridge)
            e = E_all[h].contiguous()
            packed = packed_all[h] if packed_all is not None else L.pack_codebook(e)
            xh, xst, prenorm = xs[h], x_stats[h], input_normalized
            if self.use_cosine_sim and not prenorm and not self.affine_param and (L.screen_supported(xh, C) or L.wide_dim(xh.shape[-1])):
                # cosine through the screened search (csrc/vq_screen.hip), which takes unit-norm rows: normalise once with
                # the arithmetic the exact kernel applies internally (vqp.py:37-38 at :1159), then search / sum those rows
                xh = xst = L.l2norm_rows(xh)
                prenorm = True
            # the statistics pass below reads every row next to its code: when it runs on the rows that were searched, it also
            # sums the commitment loss' squared error, and the search does not re-read x for it (csrc: vq_segsum_fast_kernel)
            sq_in_stats = (want_sqerr and do_update and x_stats is xs and (not self.use_cosine_sim or prenorm) and L.stats_sqerr_supported(xh))
            if rb is not None:
                r = dict(idx=rb["idx"][h], q=None if rb["q"] is None else rb["q"][h], sqerr_partials=None, nblk=0,
                         rnorm=None if rb["rnorm"] is None else rb["rnorm"][h])
            else:
                r = L.assign(xh, packed, e, cosine=self.use_cosine_sim, want_q=want_q, want_sqerr=want_sqerr and not sq_in_stats,
                             row_mask=rmask, skip_l2norm=prenorm, want_rnorm=self.use_cosine_sim and not prenorm,
                             q_out=q_out if H == 1 else None)
            if do_update:
                buf = torch.zeros(C * self.dim + C, dtype=torch.float32, device=x.device)
                esum, count = buf[: C * self.dim].view(C, self.dim), buf[C * self.dim:]
                if sq_in_stats:
                    _, _, parts = L.ema_accumulate(xst, r["idx"].reshape(-1), C, row_mask=rmask, count=count, embed_sum=esum,
                                                   sqerr_from=(packed, e))
                    r["sqerr_partials"], r["nblk"] = parts, parts.numel()
                else:
                    L.ema_accumulate(xst, r["idx"].reshape(-1), C, cosine=self.use_cosine_sim and not prenorm,
                                     rnorm=r["rnorm"], row_mask=rmask, count=count, embed_sum=esum)
                if self.use_ddp:
                    fused_stats_allreduce(esum, count)   # ONE collective for count || embed_sum (RCCL over xGMI)
                self._fold_stats(h, count, esum, ema_update_weight, accum_ema_update, ema_update)
            outs.append(r)
        if do_update and not accum_ema_update:
            self.expire_codes_(xs_raw.reshape(H, -1, self.dim), seq_mask=None if rmask is None else rmask[None].expand(H, -1).bool())
        if H == 1:
            return outs[0]
        if rb is not None:
            q_all, idx_all = rb["q"], rb["idx"]
        else:
            q_all, idx_all = (torch.stack([o["q"] for o in outs]) if want_q else None), torch.stack([o["idx"] for o in outs])
        return dict(q=q_all, idx=idx_all,
                    sqerr_partials=None if not want_sqerr else torch.cat([o["sqerr_partials"][: o["nblk"]] for o in outs]),
                    nblk=sum(o["nblk"] for o in outs), rnorm=None)

    @torch.no_grad()
    def update_indices(self, x: Tensor, embed_ind: Tensor, mask: Optional[Tensor] = None,
                       ema_update_weight=None, accum_ema_update=False, ema_update=None):
        """EMA update from externally supplied indices (vqp.py:643-671); -1 is treated as code 0 like
        the reference's masked_fill (:665).  x is the already-transformed input [b, n, d]."""
        ema_update = self.ema_update if ema_update is None else ema_update
        if not ema_update and not self.has_dead_code_replacement:
            return
        H, C = self.num_codebooks, self.codebook_size
        xs = x if x.ndim == 4 else x[None]
        ind = embed_ind if embed_ind.ndim == xs.ndim - 1 else embed_ind[None]
        ind = ind.masked_fill(ind == -1, 0).reshape(H, -1).contiguous()
        rmask = None if mask is None else mask.reshape(-1)
        xs_raw = xs
        if self.affine_param:        # vqp.py:594-597: the statistics are taken of the rows mapped onto the codebook's moments
            cstd = self.codebook_variance.clamp(min=1e-5).sqrt()
            bstd = self.batch_variance.clamp(min=1e-5).sqrt()
            xs = ((xs.reshape(H, -1, self.dim).float() - self.batch_mean) * (cstd / bstd) + self.codebook_mean).reshape(xs.shape)
        for h in range(H):
            buf = torch.zeros(C * self.dim + C, dtype=torch.float32, device=xs.device)
            esum, count = buf[: C * self.dim].view(C, self.dim), buf[C * self.dim:]
            L.ema_accumulate(xs[h], ind[h], C, row_mask=rmask, count=count, embed_sum=esum)
            if self.use_ddp:
                fused_stats_allreduce(esum, count)    # ONE collective for count || embed_sum
            self._fold_stats(h, count, esum, ema_update_weight, accum_ema_update, ema_update)
        if not accum_ema_update:
            self.expire_codes_(xs_raw.reshape(H, -1, self.dim), seq_mask=None if rmask is None else rmask[None].expand(H, -1).bool())

    update_ema_indices = update_indices

    def affine_codes(self, embed: Tensor) -> Tensor:
        """the codebook as it is searched under affine_param (vqp.py:721-724): mapped from its own running moments onto the batch's;
        differentiable in `embed` (the moments are buffers)"""
        cstd = self.codebook_variance.clamp(min=1e-5).sqrt()
        bstd = self.batch_variance.clamp(min=1e-5).sqrt()
        return (embed - self.codebook_mean) * (bstd / cstd) + self.batch_mean

    def forward(self, x, sample_codebook_temp=None, mask=None, freeze_codebook=False, codebook_transform_fn=None,
                ema_update_weight=None, accum_ema_update=False, ema_update=None, topk=None, update_usage=True):
        """Reference call shape (vqp.py:673-686): x is the *transformed* input; returns
        (quantize fp32, embed_ind, None) -- the N x C `dist` tensor is never produced."""
        if topk is not None:
            raise NotImplementedError("Codebook.forward(topk=): use VectorQuantize.forward(topk=) (vqhip_topk)")
        if codebook_transform_fn is not None:
            # QINCo (vqp.py:729-738, 769-776): one codebook per row; search = vqhip_assign_rowwise, quantize = a differentiable
            # gather of the transformed codes, the EMA statistics (if any) go to the base codebook as usual (:783-784)
            if self.num_codebooks != 1 or self.affine_param:
                raise NotImplementedError("codebook_transform_fn: one codebook, no affine_param")
            xf = x.float()
            if not self._is_initted():
                self.init_embed_(xf.detach().reshape(1, -1, self.dim), None if mask is None else mask.reshape(1, -1))
            embed = self.embed if self.learnable_codebook else self.embed.detach()
            if self.vq_bridge is not None:
                embed = self.vq_bridge(embed)
            te = codebook_transform_fn(embed)                                   # [1, b, n, c, d]
            te = te.reshape(*xf.shape[:-1], te.shape[-2], te.shape[-1])
            if self.use_cosine_sim:
                te = torch.nn.functional.normalize(te, p=2, dim=-1, eps=1e-6)
            ind = L.assign_rowwise(xf, te, cosine=self.use_cosine_sim)
            q = te.gather(-2, ind[..., None, None].expand(*ind.shape, 1, te.shape[-1]))[..., 0, :]
            if self.training and update_usage and not freeze_codebook:
                self.update_indices(xf.detach(), ind, mask=mask, ema_update_weight=ema_update_weight,
                                    accum_ema_update=accum_ema_update, ema_update=ema_update)
            return q, ind, None
        r = self.quantize(x.float(), mask=mask, freeze_codebook=freeze_codebook, ema_update_weight=ema_update_weight,
                          accum_ema_update=accum_ema_update, ema_update=ema_update, update_usage=update_usage,
                          input_normalized=True)
        return r["q"], r["idx"], None
