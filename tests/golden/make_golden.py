"""Generates tests/golden/*.npz from the LIVE reference (lucidrains/vector-quantize-pytorch v1.31.0,
mounted read-only at /root/reference in the build container; it does not exist on the GPU box).

    python tests/golden/make_golden.py

The reference hard-imports `einx`, which is not installed; oracle/refshim/einx provides the handful of
patterns the VQ / RVQ path uses (see its docstring).  Each fixture stores: constructor kwargs, the
state_dict before, the inputs, the reference outputs (indices always; quantized as a sha1 + float64
sum because it is exactly embed_before[indices]) and the state after.  RNG-dependent steps (k-means
seeding, dead-code replacement) are made deterministic through the reference's own seams
`Codebook.sample_fn` / `.replace_sample_fn` (vqp.py:408-410): "take the first `num` rows".
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "refshim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np
import torch

from vector_quantize_pytorch import (GroupedResidualVQ, HierarchicalVQ, RandomProjectionQuantizer, ResidualSimVQ, ResidualVQ, SimVQ,
                                     VectorQuantize)  # the live reference


def first_rows(samples, num):           # deterministic stand-in for batched_sample_vectors (vqp.py:165)
    n = samples.shape[1]
    if n >= num:
        return samples[:, :num].clone()
    reps = -(-num // n)
    return samples.repeat(1, reps, 1)[:, :num].clone()


def sha(t):
    return hashlib.sha1(to_np(t.detach().contiguous()).tobytes()).hexdigest()


def to_np(t):
    t = t.detach().cpu().clone()          # clone: state_dict tensors alias live buffers that the forward mutates
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy()          # raw bits; loader views them back
    return t.numpy()


def inject(mod):
    for m in mod.modules():
        if hasattr(m, "sample_fn"):
            m.sample_fn = first_rows
            m.replace_sample_fn = first_rows


ONLY = set(sys.argv[1:])       # python tests/golden/make_golden.py [name ...]: (re)generate just these fixtures


def run_case(name, cls, kwargs, xs, *, train=True, fwd_kwargs=None, grad=False, unit_codebook=False, deterministic_sampling=False,
             build=None, param_grad=False):
    if ONLY and name not in ONLY:
        return
    torch.manual_seed(1234)
    mod = cls(**kwargs) if build is None else build()
    if unit_codebook:
        for m in mod.modules():
            if hasattr(m, "embed"):
                e = torch.randn_like(m.embed)
                if getattr(m, "use_cosine_sim", False):
                    e = torch.nn.functional.normalize(e, dim=-1)
                m.embed.data.copy_(e)
                m.embed_avg.data.copy_(e)
    if deterministic_sampling:
        inject(mod)
    mod.train(train)
    out = {"meta": dict(name=name, cls=cls.__name__, kwargs={k: (list(v) if isinstance(v, tuple) else v) for k, v in kwargs.items()},
                        train=train, steps=len(xs), grad=grad, fwd_kwargs=fwd_kwargs or {},
                        deterministic_sampling=deterministic_sampling, bf16=bool(xs[0].dtype == torch.bfloat16),
                        build=None if build is None else name, param_grad=param_grad)}
    arrays = {}
    shared = bool(kwargs.get("shared_codebook", False))

    def aliased(k):     # shared_codebook: layers.{i>0} alias layers.0 (rvq.py:302-306); store once
        return shared and k.startswith("layers.") and not k.startswith("layers.0.")

    for k, v in mod.state_dict().items():
        if aliased(k):
            continue
        if k.endswith("embed_avg") and torch.equal(v, mod.state_dict()[k[:-4]]):
            continue                                 # loader: missing embed_avg == embed
        arrays["before/" + k] = to_np(v)
    fk = dict(fwd_kwargs or {})
    if "lens" in fk:
        fk["lens"] = torch.tensor(fk["lens"])
    if "indices" in fk:
        fk["indices"] = torch.tensor(fk["indices"])
    if "mask" in fk:
        fk["mask"] = torch.tensor(fk["mask"])
    for s, x in enumerate(xs):
        x = x.clone()
        if grad:
            x.requires_grad_(True)
        res = mod(x, **fk)
        if torch.is_tensor(res) and res.dtype.is_floating_point:   # RandomProjectionQuantizer(indices=) returns the cross-entropy only
            res = (torch.zeros(1), torch.zeros(1, dtype=torch.long), res)
        if torch.is_tensor(res):                      # RandomProjectionQuantizer returns indices only
            res = (torch.zeros(1), res, torch.zeros(()))
        if len(res) == 2:                             # forward(indices=...) returns (quantize, cross-entropy loss)
            res = (res[0], torch.zeros(1, dtype=torch.long), res[1])
        q, idx, loss = res[0], res[1], res[2]
        if isinstance(idx, tuple):                    # HierarchicalVQ: one index map per scale
            idx = torch.cat([i.flatten(1) for i in idx], 1)
        arrays[f"x{s}"] = to_np(x)
        arrays[f"idx{s}"] = to_np(idx)
        arrays[f"loss{s}"] = to_np(loss.float())
        arrays[f"qsum{s}"] = np.float64(q.double().sum().item())
        out["meta"][f"qsha{s}"] = sha(q)
        if grad or q.dtype != torch.float32 or "lens" in fk or cls is not VectorQuantize:
            arrays[f"q{s}"] = to_np(q)          # not reconstructible as embed[idx]: store it
        if grad or param_grad:
            g = torch.Generator().manual_seed(77 + s)
            w = torch.randn(q.shape, generator=g).to(q.dtype)
            for p_ in mod.parameters():
                p_.grad = None
            (loss.sum() * 3.0 + (q * w).sum()).backward()
            arrays[f"gw{s}"] = to_np(w)
            if grad:
                arrays[f"gx{s}"] = to_np(x.grad)
            if param_grad:
                for pn, p_ in mod.named_parameters():
                    if p_.grad is not None:
                        arrays[f"pg{s}/{pn}"] = to_np(p_.grad)
    if train:
        for k, v in mod.state_dict().items():
            if not aliased(k):
                arrays["after/" + k] = to_np(v)
    arrays["meta"] = np.frombuffer(json.dumps(out["meta"]).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB  loss0={float(arrays['loss0'].reshape(-1)[0]):.6f}")


def run_big_case(name, kwargs, shape, seed, *, grad):
    """tests/golden/big/<name>.npz: ONE training step of the live reference's ResidualVQ at a size where near-ties show up
    (>= 65 536 rows x 8 stages, default tiny kaiming codebook).  Stored: the codebook before the step, the INDICES
    (int16: codebook_size <= 32768), the per-stage losses and a sha1 of x -- x itself is re-drawn by the tests from `seed` with
    torch's CPU generator (same torch build on the GPU box; the sha1 guards that)."""
    if ONLY and name not in ONLY:
        return
    torch.manual_seed(1234)
    mod = ResidualVQ(**kwargs).train()
    before = {k: to_np(v) for k, v in mod.state_dict().items() if k.startswith("layers.0._codebook.") and not k.endswith("embed_avg")}
    x = randn(*shape, seed=seed)
    xin = x.clone().requires_grad_(True) if grad else x.clone()
    _, idx, losses = mod(xin)
    assert int(idx.max()) < 32768
    meta = dict(name=name, kwargs=kwargs, shape=list(shape), seed=seed, grad=grad, xsha=sha(x))
    os.makedirs(os.path.join(HERE, "big"), exist_ok=True)
    path = os.path.join(HERE, "big", name + ".npz")
    np.savez_compressed(path, idx=idx.numpy().astype(np.int16), losses=to_np(losses.float()),
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **{"before/" + k: v for k, v in before.items()})
    print(f"big/{name}: {os.path.getsize(path) / 1e6:.2f} MB  losses={losses.detach().numpy().round(5).tolist()}")


def randn(*shape, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


if __name__ == "__main__":
    torch.set_num_threads(8)
    # BASELINE cfg 1: README smoke test shape, default (tiny, tie-prone) kaiming codebook
    run_case("vq_cfg1_train", VectorQuantize, dict(dim=256, codebook_size=512), [randn(1, 1024, 256, seed=1)])
    run_case("vq_cfg1_eval", VectorQuantize, dict(dim=256, codebook_size=512), [randn(1, 1024, 256, seed=2)], train=False)
    run_case("vq_unit_2step", VectorQuantize, dict(dim=128, codebook_size=256), [randn(2, 300, 128, seed=3), randn(2, 300, 128, seed=4)], unit_codebook=True)
    run_case("vq_cosine", VectorQuantize, dict(dim=128, codebook_size=256, use_cosine_sim=True), [randn(1, 512, 128, seed=5), randn(1, 512, 128, seed=6)])
    # cfg 2 dtype (bf16 in), scaled down
    run_case("vq_bf16", VectorQuantize, dict(dim=256, codebook_size=1024), [randn(2, 512, 256, seed=7, dtype=torch.bfloat16)])
    run_case("vq_lens", VectorQuantize, dict(dim=64, codebook_size=128), [randn(3, 200, 64, seed=8)], fwd_kwargs=dict(lens=[200, 57, 1]), unit_codebook=True)
    run_case("vq_grad_rot", VectorQuantize, dict(dim=64, codebook_size=128), [randn(2, 128, 64, seed=9)], grad=True, unit_codebook=True)
    run_case("vq_grad_ste", VectorQuantize, dict(dim=64, codebook_size=128, rotation_trick=False), [randn(2, 128, 64, seed=10)], grad=True, unit_codebook=True)
    run_case("vq_grad_cos", VectorQuantize, dict(dim=64, codebook_size=128, use_cosine_sim=True), [randn(2, 128, 64, seed=11)], grad=True)
    run_case("vq_kmeans", VectorQuantize, dict(dim=32, codebook_size=64, kmeans_init=True, kmeans_iters=4), [randn(1, 2048, 32, seed=12)], deterministic_sampling=True)
    run_case("vq_expire", VectorQuantize, dict(dim=32, codebook_size=128, threshold_ema_dead_code=2), [randn(1, 256, 32, seed=13), randn(1, 256, 32, seed=14)],
             unit_codebook=True, deterministic_sampling=True)
    run_case("vq_fmap", VectorQuantize, dict(dim=32, codebook_size=64, accept_image_fmap=True), [randn(2, 32, 8, 8, seed=15)], unit_codebook=True)
    run_case("vq_heads", VectorQuantize, dict(dim=64, codebook_size=64, heads=4, codebook_dim=16), [randn(2, 60, 64, seed=30)], unit_codebook=True)
    run_case("vq_heads_sep", VectorQuantize, dict(dim=64, codebook_size=64, heads=4, codebook_dim=16, separate_codebook_per_head=True),
             [randn(2, 60, 64, seed=31)], unit_codebook=True)
    run_case("vq_3d", VectorQuantize, dict(dim=32, codebook_size=64, accept_3d_fmap=True), [randn(1, 32, 4, 4, 4, seed=32)], unit_codebook=True)
    run_case("vq_channel_first", VectorQuantize, dict(dim=32, codebook_size=64, channel_last=False), [randn(2, 32, 50, seed=33)], unit_codebook=True)
    # codebooks that receive gradients (search on the HIP kernel, autograd glue in torch)
    run_case("vq_learnable", VectorQuantize, dict(dim=64, codebook_size=128, learnable_codebook=True, ema_update=False),
             [randn(2, 100, 64, seed=40)], grad=True, param_grad=True, unit_codebook=True)
    run_case("vq_learnable_sync_v", VectorQuantize, dict(dim=32, codebook_size=64, learnable_codebook=True, ema_update=False, sync_update_v=0.3, rotation_trick=False),
             [randn(2, 80, 32, seed=41)], grad=True, param_grad=True, unit_codebook=True)
    run_case("vq_orthogonal", VectorQuantize, dict(dim=32, codebook_size=64, orthogonal_reg_weight=10.),
             [randn(2, 80, 32, seed=42)], param_grad=True, unit_codebook=True)
    run_case("vq_inplace_opt", VectorQuantize, dict(dim=32, codebook_size=64, learnable_codebook=True, ema_update=False),
             [randn(2, 80, 32, seed=43), randn(2, 80, 32, seed=44)], unit_codebook=True,
             build=lambda: VectorQuantize(dim=32, codebook_size=64, learnable_codebook=True, ema_update=False,
                                          in_place_codebook_optimizer=lambda p: torch.optim.SGD(p, lr=0.5)))
    run_case("vq_bridge", VectorQuantize, dict(dim=32, codebook_size=64, learnable_codebook=True, ema_update=False),
             [randn(2, 80, 32, seed=45)], grad=True, param_grad=True, unit_codebook=True,
             build=lambda: VectorQuantize(dim=32, codebook_size=64, vq_bridge=torch.nn.Linear(32, 32)))
    # SimVQ / ResidualSimVQ (torch.cdist + argmin on an implicit codebook)
    run_case("simvq", SimVQ, dict(dim=64, codebook_size=256), [randn(2, 150, 64, seed=50)], grad=True, param_grad=True)
    run_case("simvq_ste_channel_first", SimVQ, dict(dim=32, codebook_size=128, rotation_trick=False, channel_first=True),
             [randn(2, 32, 6, 6, seed=51)], grad=True, param_grad=True)
    run_case("residual_simvq", ResidualSimVQ, dict(dim=64, num_quantizers=4, codebook_size=128), [randn(2, 100, 64, seed=52)], grad=True, param_grad=True)
    # callers
    run_case("rpq", RandomProjectionQuantizer, dict(dim=64, codebook_size=32, codebook_dim=16, num_codebooks=4), [randn(2, 50, 64, seed=60)])
    run_case("rpq_indices", RandomProjectionQuantizer, dict(dim=64, codebook_size=32, codebook_dim=16, num_codebooks=4), [randn(2, 50, 64, seed=64)],
             fwd_kwargs=dict(indices=torch.randint(0, 32, (2, 50, 4), generator=torch.Generator().manual_seed(6)).tolist()), grad=True)
    run_case("hvq", HierarchicalVQ, dict(dim=32, codebook_size=64, scales=(1, 2, 4, 8), accept_image_fmap=True),
             [randn(2, 32, 8, 8, seed=61), randn(2, 32, 8, 8, seed=62)], deterministic_sampling=True)
    run_case("hvq_nokmeans", HierarchicalVQ, dict(dim=32, codebook_size=64, scales=(2, 4), kmeans_init=False, threshold_ema_dead_code=0,
                                                  rotation_trick=True, share_quant_resi=2, accept_image_fmap=True),
             [randn(2, 32, 8, 8, seed=63)], grad=True, unit_codebook=True)
    # options that read the whole distance row
    run_case("vq_ce_commit", VectorQuantize, dict(dim=32, codebook_size=64, commitment_use_cross_entropy_loss=True),
             [randn(2, 80, 32, seed=70)], grad=True, unit_codebook=True)
    run_case("vq_diversity", VectorQuantize, dict(dim=32, codebook_size=64, codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=10.),
             [randn(2, 80, 32, seed=71)], grad=True, unit_codebook=True)
    run_case("vq_topk", VectorQuantize, dict(dim=32, codebook_size=64), [randn(2, 40, 32, seed=72)], fwd_kwargs=dict(topk=3), unit_codebook=True)
    run_case("vq_topk_cos", VectorQuantize, dict(dim=32, codebook_size=64, use_cosine_sim=True), [randn(2, 40, 32, seed=73)],
             fwd_kwargs=dict(topk=2))       # (the reference's eval-mode gather does not support topk)
    run_case("vq_indices_ce", VectorQuantize, dict(dim=32, codebook_size=64), [randn(2, 40, 32, seed=74)],
             fwd_kwargs=dict(indices=torch.randint(0, 64, (2, 40), generator=torch.Generator().manual_seed(5)).tolist()), grad=True, unit_codebook=True)
    run_case("vq_stochastic_temp0", VectorQuantize, dict(dim=32, codebook_size=64, stochastic_sample_codes=True, sample_codebook_temp=0.),
             [randn(2, 80, 32, seed=75)], unit_codebook=True)
    run_case("vq_gumbel_st", VectorQuantize, dict(dim=32, codebook_size=64, straight_through=True, rotation_trick=False, sample_codebook_temp=0.5),
             [randn(2, 80, 32, seed=76)], grad=True, unit_codebook=True)
    # beam search: batch 1 only -- the reference's post-search update_indices mis-shapes the indices for batch > 1 (vqp.py:664)
    run_case("rvq_beam", ResidualVQ, dict(dim=32, num_quantizers=4, codebook_size=64, beam_size=3), [randn(1, 80, 32, seed=80)], unit_codebook=True)
    run_case("rvq_beam_shared_mask", ResidualVQ, dict(dim=32, num_quantizers=3, codebook_size=64, beam_size=2, shared_codebook=True,
                                                      beam_score_quantizer_weights=[1., 0.5, 0.25]),
             [randn(1, 60, 32, seed=81)], fwd_kwargs=dict(mask=[[True] * 43 + [False] * 17]), unit_codebook=True)
    run_case("vq_affine", VectorQuantize, dict(dim=32, codebook_size=64, affine_param=True, affine_param_batch_decay=0.9, affine_param_codebook_decay=0.8),
             [randn(2, 80, 32, seed=90) * 2 + 1, randn(2, 80, 32, seed=91) * 2 + 1], unit_codebook=True)
    run_case("vq_proj", VectorQuantize, dict(dim=48, codebook_size=64, codebook_dim=16), [randn(2, 50, 48, seed=16)], unit_codebook=True)
    # cfg 3: ResidualVQ shared codebook, scaled down
    run_case("rvq_shared", ResidualVQ, dict(dim=256, num_quantizers=8, codebook_size=256, shared_codebook=True), [randn(2, 128, 256, seed=17), randn(2, 128, 256, seed=18)])
    run_case("rvq_separate", ResidualVQ, dict(dim=64, num_quantizers=4, codebook_size=128), [randn(2, 200, 64, seed=19)], unit_codebook=True)
    run_case("rvq_tiger", ResidualVQ, dict(dim=2, codebook_size=(5, 128, 256)), [randn(2, 32, 2, seed=20)], unit_codebook=True)
    run_case("rvq_cosine_eval", ResidualVQ, dict(dim=64, num_quantizers=3, codebook_size=64, use_cosine_sim=True), [randn(1, 256, 64, seed=21)], train=False)
    # the residual loop with bf16 rows (every tensor op of the reference rounds to bf16: the residual update, the running sum) and
    # with cosine codebooks in training (EMA on l2-normalised rows, staged path)
    run_case("rvq_bf16", ResidualVQ, dict(dim=64, num_quantizers=3, codebook_size=64), [randn(2, 120, 64, seed=120, dtype=torch.bfloat16),
             randn(2, 120, 64, seed=121, dtype=torch.bfloat16)], unit_codebook=True)
    run_case("rvq_cosine_train", ResidualVQ, dict(dim=64, num_quantizers=3, codebook_size=64, use_cosine_sim=True),
             [randn(2, 150, 64, seed=122), randn(2, 150, 64, seed=123)], unit_codebook=True)
    # quantizer dropout (rvq.py:478-482, seeded through rand_quantize_dropout_fixed_seed): dropped stages return -1 / zero loss
    run_case("rvq_dropout", ResidualVQ, dict(dim=32, num_quantizers=4, codebook_size=64, quantize_dropout=True, quantize_dropout_cutoff_index=1),
             [randn(2, 90, 32, seed=124), randn(2, 90, 32, seed=125)], fwd_kwargs=dict(rand_quantize_dropout_fixed_seed=3), unit_codebook=True)
    # cosine codebook, variable lengths, training without an input gradient: the masked commitment loss compares against the
    # un-normalised input (vqp.py:1319) while the EMA statistics see the normalised rows
    run_case("vq_cosine_lens_train", VectorQuantize, dict(dim=32, codebook_size=64, use_cosine_sim=True),
             [randn(3, 40, 32, seed=126) * 2.0, randn(3, 40, 32, seed=127) * 2.0], fwd_kwargs=dict(lens=[40, 17, 29]), unit_codebook=True)
    # bf16 rows with the cosine metric (the reference l2-normalises in bf16: norm and quotient rounded), k-means init with the cosine
    # metric (means re-normalised every iteration, vqp.py:262-276), grouped residual VQ on bf16 rows
    run_case("vq_bf16_cos", VectorQuantize, dict(dim=64, codebook_size=128, use_cosine_sim=True),
             [randn(2, 256, 64, seed=130, dtype=torch.bfloat16), randn(2, 256, 64, seed=131, dtype=torch.bfloat16)])
    run_case("vq_kmeans_cos", VectorQuantize, dict(dim=32, codebook_size=32, use_cosine_sim=True, kmeans_init=True, kmeans_iters=4),
             [randn(1, 1024, 32, seed=132), randn(1, 1024, 32, seed=133)], deterministic_sampling=True)
    run_case("grvq_bf16", GroupedResidualVQ, dict(dim=128, groups=2, num_quantizers=3, codebook_size=64),
             [randn(2, 100, 128, seed=134, dtype=torch.bfloat16)], unit_codebook=True)
    # dead-code replacement inside the residual loop (separate codebooks: every layer replaces its expired codes with rows of ITS stage
    # input, vqp.py:564-574 through the deterministic sampler), and two EMA steps on bf16 rows
    run_case("rvq_expire", ResidualVQ, dict(dim=32, num_quantizers=3, codebook_size=128, threshold_ema_dead_code=2),
             [randn(1, 256, 32, seed=140), randn(1, 256, 32, seed=141)], unit_codebook=True, deterministic_sampling=True)
    run_case("vq_bf16_2step", VectorQuantize, dict(dim=64, codebook_size=256),
             [randn(2, 300, 64, seed=142, dtype=torch.bfloat16), randn(2, 300, 64, seed=143, dtype=torch.bfloat16)], unit_codebook=True)
    # cfg 5: grouped RVQ, scaled down (k-means through the deterministic sampler)
    run_case("grvq", GroupedResidualVQ, dict(dim=128, groups=2, num_quantizers=3, codebook_size=64), [randn(2, 100, 128, seed=22)], unit_codebook=True)
    run_case("grvq_kmeans", GroupedResidualVQ, dict(dim=64, groups=2, num_quantizers=2, codebook_size=32, kmeans_init=True, kmeans_iters=3),
             [randn(1, 1024, 64, seed=23)], deterministic_sampling=True)
    # QINCo (rvq.py:107-162, 288-289, 460-499): implicit neural codebooks -- every quantizer after the first searches a codebook
    # that an MLP derives per row from the sum so far; learnable codebooks, gradients to the codes, the MLPs and the input
    qinco = dict(dim=32, num_quantizers=3, codebook_size=64, implicit_neural_codebook=True, mlp_kwargs=dict(depth=2))
    run_case("rvq_qinco", ResidualVQ, qinco, [randn(2, 50, 32, seed=100)], grad=True, param_grad=True, unit_codebook=True)
    run_case("rvq_qinco_eval", ResidualVQ, qinco, [randn(2, 40, 32, seed=101)], train=False, unit_codebook=True)
    # the per-row codebook path with the COSINE metric and an input that does not require grad (the reference l2-normalises
    # x before anything else, vqp.py:1159, so the commitment loss and the EMA statistics see normalised rows).  QINCo itself cannot be
    # cosine -- the reference asserts against cosine + learnable_codebook (vqp.py:884) -- so the fixture drives codebook_transform_fn
    # directly, through the harness shared with the tests
    import importlib.util
    _spec = importlib.util.spec_from_file_location("golden_util", os.path.join(ROOT, "tests", "golden_util.py"))
    _gu = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_gu)
    run_case("vq_cos_transform_nograd", VectorQuantize, dict(dim=32, codebook_size=64, use_cosine_sim=True),
             [randn(2, 50, 32, seed=102) * 3.0 + 0.5, randn(2, 50, 32, seed=103) * 3.0 + 0.5], unit_codebook=True,
             build=lambda: _gu.TransformCaller(VectorQuantize(dim=32, codebook_size=64, use_cosine_sim=True)))
    # gradients to the input through the residual loop (rvq.py:524-525 with quant_grad_frac = 0): every stage's rotation-trick
    # (the default, vqp.py:856) / straight-through Jacobian (vqp.py:1225-1233) plus every stage's commitment-loss gradient, summed into dL/dx
    run_case("rvq_shared_grad", ResidualVQ, dict(dim=64, num_quantizers=4, codebook_size=128, shared_codebook=True),
             [randn(2, 128, 64, seed=110), randn(2, 128, 64, seed=111)], grad=True, unit_codebook=True)
    run_case("rvq_grad_ste", ResidualVQ, dict(dim=64, num_quantizers=3, codebook_size=64, rotation_trick=False),
             [randn(2, 100, 64, seed=112)], grad=True, unit_codebook=True)
    run_case("rvq_grad_mask", ResidualVQ, dict(dim=32, num_quantizers=3, codebook_size=64),
             [randn(2, 60, 32, seed=113)], fwd_kwargs=dict(mask=[[True] * 60, [True] * 37 + [False] * 23]), grad=True, unit_codebook=True)
    run_case("grvq_grad", GroupedResidualVQ, dict(dim=128, groups=2, num_quantizers=3, codebook_size=64),
             [randn(2, 100, 128, seed=114)], grad=True, unit_codebook=True)
    # cross-entropy commitment on the FIRST step of a k-means-initialised codebook -- the reference initialises before it
    # computes `dist` (vqp.py:718-720), so the loss is taken against the initialised codes, not the all-zero buffer
    run_case("vq_ce_kmeans", VectorQuantize, dict(dim=32, codebook_size=32, kmeans_init=True, kmeans_iters=3, commitment_use_cross_entropy_loss=True),
             [randn(2, 400, 32, seed=150), randn(2, 400, 32, seed=151)], grad=True, deterministic_sampling=True)
    # several heads together with the options that read whole score rows, affine_param, learnable codebooks -- the
    # reference carries the head axis through every einsum (vqp.py:1044-1049, 1242-1292): one shared codebook on [(b h), n, d] rows, or
    # one codebook per head.  (topk with heads > 1 fails upstream: rearrange '1 (b h) n -> b n h' of a [1, b h, n, k] tensor.)
    hd = dict(dim=64, codebook_size=64, heads=4, codebook_dim=16)
    run_case("vq_heads_ce", VectorQuantize, dict(hd, commitment_use_cross_entropy_loss=True), [randn(2, 60, 64, seed=160)], grad=True, unit_codebook=True)
    run_case("vq_heads_diversity", VectorQuantize, dict(hd, codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=10.),
             [randn(2, 60, 64, seed=161)], grad=True, unit_codebook=True)
    run_case("vq_heads_gumbel_st", VectorQuantize, dict(dim=32, codebook_size=64, heads=2, codebook_dim=16, straight_through=True, rotation_trick=False,
                                                        sample_codebook_temp=0.5), [randn(2, 60, 32, seed=162)], grad=True, unit_codebook=True)
    run_case("vq_heads_affine", VectorQuantize, dict(hd, affine_param=True, affine_param_batch_decay=0.9, affine_param_codebook_decay=0.8),
             [randn(2, 60, 64, seed=163) * 2 + 1, randn(2, 60, 64, seed=164) * 2 + 1], unit_codebook=True)
    hs = dict(hd, separate_codebook_per_head=True)
    run_case("vq_heads_sep_ce", VectorQuantize, dict(hs, commitment_use_cross_entropy_loss=True), [randn(2, 60, 64, seed=165)], grad=True, unit_codebook=True)
    run_case("vq_heads_sep_diversity", VectorQuantize, dict(hs, codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=10.),
             [randn(2, 60, 64, seed=166)], grad=True, unit_codebook=True)
    run_case("vq_heads_sep_learnable", VectorQuantize, dict(hs, learnable_codebook=True, ema_update=False),
             [randn(2, 60, 64, seed=167)], grad=True, param_grad=True, unit_codebook=True)
    run_case("vq_heads_sep_affine", VectorQuantize, dict(hs, affine_param=True), [randn(2, 60, 64, seed=168) * 2 + 1, randn(2, 60, 64, seed=169) * 2 + 1],
             unit_codebook=True)
    # codebook dims beyond 512 (the reference takes any dim, vqp.py:803-806): 768 and 1024 -- the ATen sum of squares
    # of a row gets its second cascade level there -- train steps with the rotation trick, cosine, bf16 rows, a residual VQ
    run_case("vq_dim1024", VectorQuantize, dict(dim=1024, codebook_size=512), [randn(1, 160, 1024, seed=170)])
    run_case("vq_dim768_grad", VectorQuantize, dict(dim=768, codebook_size=64), [randn(2, 60, 768, seed=172)], grad=True, unit_codebook=True)
    run_case("vq_dim1024_cos", VectorQuantize, dict(dim=1024, codebook_size=64, use_cosine_sim=True),
             [randn(1, 100, 1024, seed=173), randn(1, 100, 1024, seed=174)])
    run_case("vq_dim640_bf16", VectorQuantize, dict(dim=640, codebook_size=64), [randn(2, 100, 640, seed=175, dtype=torch.bfloat16)], unit_codebook=True)
    run_case("rvq_dim768", ResidualVQ, dict(dim=768, num_quantizers=3, codebook_size=48), [randn(1, 100, 768, seed=176)], grad=True, unit_codebook=True)
    run_case("vq_dim2048_eval", VectorQuantize, dict(dim=2048, codebook_size=64), [randn(1, 70, 2048, seed=177)], train=False)
    # Regressions found by the random option combinations of make_combo.py (round 5), each reduced to a named case:
    # * k-means init of one codebook per head next to the streamed cross-entropy, padded batch (the init got [1, (h b n), d] rows)
    run_case("vq_heads_sep_ce_kmeans_lens", VectorQuantize, dict(dim=64, codebook_size=32, heads=2, codebook_dim=8, separate_codebook_per_head=True,
                                                                  kmeans_init=True, kmeans_iters=3, commitment_use_cross_entropy_loss=True),
             [randn(2, 300, 64, seed=180)], fwd_kwargs=dict(lens=[300, 210]), grad=True, deterministic_sampling=True)
    # * affine_param together with a learnable codebook (the parameter's gradient passes through the map onto the batch's moments,
    #   vqp.py:721-724) and with the options that read whole score rows (the scores are those of the MAPPED codebook)
    af = dict(dim=32, codebook_size=64, affine_param=True)
    run_case("vq_affine_learnable", VectorQuantize, dict(af, learnable_codebook=True, ema_update=False),
             [randn(2, 80, 32, seed=181) * 2 + 1, randn(2, 80, 32, seed=182) * 2 + 1], grad=True, param_grad=True, unit_codebook=True)
    run_case("vq_affine_ce", VectorQuantize, dict(af, commitment_use_cross_entropy_loss=True),
             [randn(2, 80, 32, seed=183) * 2 + 1, randn(2, 80, 32, seed=184) * 2 + 1], grad=True, unit_codebook=True)
    run_case("vq_affine_diversity", VectorQuantize, dict(af, codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=10.),
             [randn(2, 80, 32, seed=185) * 2 + 1, randn(2, 80, 32, seed=186) * 2 + 1], grad=True, unit_codebook=True)
    run_case("vq_affine_topk", VectorQuantize, af, [randn(2, 40, 32, seed=187) * 2 + 1], fwd_kwargs=dict(topk=3), unit_codebook=True)
    run_case("vq_affine_learnable_ce_lens", VectorQuantize, dict(af, learnable_codebook=True, ema_update=False, commitment_use_cross_entropy_loss=True),
             [randn(2, 80, 32, seed=188) * 2 + 1], fwd_kwargs=dict(lens=[80, 31]), grad=True, param_grad=True, unit_codebook=True)
    # * the masked commitment loss of a multi-headed module: the reference compares every head's codes with the CALLER's tensor
    #   (`orig_input`, vqp.py:1108, 1319) -- shapes that only broadcast when codebook_dim == dim (it raises otherwise)
    run_case("vq_heads_sep_mask_origdim", VectorQuantize, dict(dim=32, codebook_size=32, heads=2, codebook_dim=32, separate_codebook_per_head=True,
                                                                rotation_trick=False),
             [randn(2, 100, 32, seed=189), randn(2, 100, 32, seed=190)], fwd_kwargs=dict(lens=[100, 41]), unit_codebook=True)
    run_case("vq_heads_mask_origdim", VectorQuantize, dict(dim=32, codebook_size=32, heads=4, codebook_dim=32, affine_param=True),
             [randn(1, 90, 32, seed=191)], fwd_kwargs=dict(lens=[33]), unit_codebook=True)
    # * bf16 rows that require grad: rotate_to runs on bf16 TENSORS (vqp.py:287-318), every op rounds -- ~2 % away from the fp32 formula,
    #   and the residual loop subtracts exactly that value (rvq.py:524): later stages' indices depend on it
    bf = torch.bfloat16
    run_case("vq_bf16_grad_rot", VectorQuantize, dict(dim=64, codebook_size=128), [randn(2, 200, 64, seed=192, dtype=bf)], grad=True, unit_codebook=True)
    run_case("vq_bf16_grad_rot_lens", VectorQuantize, dict(dim=32, codebook_size=64), [randn(2, 200, 32, seed=193, dtype=bf)],
             fwd_kwargs=dict(lens=[200, 77]), grad=True, unit_codebook=True)
    run_case("vq_bf16_grad_ste", VectorQuantize, dict(dim=64, codebook_size=128, rotation_trick=False), [randn(2, 200, 64, seed=194, dtype=bf)],
             grad=True, unit_codebook=True)
    run_case("vq_bf16_cos_grad_rot", VectorQuantize, dict(dim=64, codebook_size=128, use_cosine_sim=True), [randn(2, 200, 64, seed=195, dtype=bf)], grad=True)
    run_case("rvq_bf16_grad_rot", ResidualVQ, dict(dim=64, num_quantizers=4, codebook_size=64, shared_codebook=True),
             [randn(2, 300, 64, seed=196, dtype=bf), randn(2, 300, 64, seed=197, dtype=bf)], grad=True, unit_codebook=True)
    run_case("rvq_bf16_grad_ste", ResidualVQ, dict(dim=64, num_quantizers=4, codebook_size=64, rotation_trick=False),
             [randn(2, 300, 64, seed=198, dtype=bf)], grad=True, unit_codebook=True)
    run_case("grvq_bf16_grad_rot", GroupedResidualVQ, dict(dim=128, groups=2, num_quantizers=3, codebook_size=64),
             [randn(2, 200, 128, seed=199, dtype=bf)], grad=True, unit_codebook=True)
    run_case("vq_bf16_grad_rot_d256", VectorQuantize, dict(dim=256, codebook_size=256), [randn(2, 150, 256, seed=201, dtype=bf)], grad=True, unit_codebook=True)
    # the same loop at a size where near-ties show up: 65 536 rows x 8 stages x 1024 shared codes, default init (cfg 3's shape, a
    # quarter of its rows) -- without an input gradient, and with one under the rotation trick (default) / straight-through, where
    # rvq.py:524 subtracts the layer's ROUTED value from the residual and the later stages' indices depend on its last bits
    big = dict(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True)
    run_big_case("rvq_big_nograd", big, (8, 8192, 256), 200, grad=False)
    run_big_case("rvq_big_rot", big, (8, 8192, 256), 200, grad=True)
    run_big_case("rvq_big_ste", dict(big, rotation_trick=False), (8, 8192, 256), 200, grad=True)
