"""Generates tests/golden/combo_*.npz: a seeded random walk over COMBINATIONS of the modules' options, each one step or two of the
LIVE reference (same recipe and file format as make_golden.py, whose `run_case` does the work).

    python tests/golden/make_combo.py [first_seed [count]]

The hand-written fixtures of make_golden.py cover every option once; this file covers options TOGETHER (projection x heads x metric
x masks x dead codes x losses that read the score row x gradients x layouts x dtypes x residual loops), which is where a drop-in
breaks first.  A draw the reference itself rejects (its constructor asserts, or its forward fails) is skipped and printed.  Draws
avoid the reference's RNG-dependent branches that have no seam (gumbel noise at temperature > 0, orthogonal_reg_max_codes'
randperm); k-means seeding and dead-code replacement go through `sample_fn` / `replace_sample_fn` like in make_golden.py.
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import torch

import make_golden as M
from make_golden import GroupedResidualVQ, ResidualVQ, VectorQuantize, randn


def draw_vq(r: random.Random, seed: int):
    heads = r.choice([1, 1, 1, 2, 4])
    dim = r.choice([16, 32, 40, 64, 96, 128])
    kw = dict(dim=dim, codebook_size=r.choice([17, 32, 64, 100, 128, 300]))
    if heads > 1:
        kw.update(heads=heads, codebook_dim=r.choice([8, 16, 32]))
        if r.random() < 0.5:
            kw["separate_codebook_per_head"] = True
    elif r.random() < 0.3:
        kw["codebook_dim"] = r.choice([8, 16, 24])
        if r.random() < 0.3:
            kw["layernorm_after_project_in"] = True
    cosine = r.random() < 0.3
    learnable = (not cosine) and r.random() < 0.2
    if cosine:
        kw["use_cosine_sim"] = True
    if learnable:
        kw.update(learnable_codebook=True, ema_update=False)
        if r.random() < 0.5:
            kw.update(sync_update_v=r.choice([0.1, 0.5]), rotation_trick=False)
    if "rotation_trick" not in kw and r.random() < 0.4:
        kw["rotation_trick"] = False
    if r.random() < 0.4:
        kw["commitment_weight"] = r.choice([0.25, 2.0])
    if r.random() < 0.4:
        kw["decay"] = r.choice([0.5, 0.99])
    det = False
    if not learnable and r.random() < 0.25:
        kw["threshold_ema_dead_code"] = 2
        det = True
    if r.random() < 0.15:
        kw.update(kmeans_init=True, kmeans_iters=3)
        det = True
    extra = r.random()
    if extra < 0.12:
        kw["commitment_use_cross_entropy_loss"] = True
    elif extra < 0.24:
        kw.update(codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=r.choice([10., 100.]))
    elif extra < 0.36:
        kw.update(orthogonal_reg_weight=5., orthogonal_reg_active_codes_only=r.random() < 0.5)
    elif extra < 0.44 and not cosine:
        kw["affine_param"] = True
    elif extra < 0.50:
        kw.update(stochastic_sample_codes=True, sample_codebook_temp=0.)
    b, n = r.choice([1, 2, 3]), r.choice([33, 64, 100, 150])
    if kw.get("kmeans_init") or det:
        # k-means seeding / dead-code replacement on fewer rows than a few per code fill the codebook with (near-)duplicates of the
        # batch rows: every later search is then decided by ties in the last bit, which no two BLAS builds agree on either
        kw["codebook_size"] = min(kw["codebook_size"], 64)
        n = max(n, -(-12 * kw["codebook_size"] // b))
    steps = r.choice([1, 2])
    layout = r.random()
    bf16 = r.random() < 0.12 and extra >= 0.50 and not learnable and not (cosine and det)
    dtype = torch.bfloat16 if bf16 else torch.float32
    fwd = {}
    if layout < 0.12:
        side = r.choice([5, 8])
        kw["accept_image_fmap"] = True
        xs = [randn(b, dim, side, side, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
    elif layout < 0.24:
        kw["channel_last"] = False
        xs = [randn(b, dim, n, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
    else:
        xs = [randn(b, n, dim, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
        m = r.random()
        if m < 0.2:
            fwd["lens"] = [r.randint(1, n) for _ in range(b)]
        elif m < 0.35:
            fwd["mask"] = [[r.random() < 0.7 for _ in range(n)] for _ in range(b)]
    if r.random() < 0.1:
        fwd["freeze_codebook"] = True
    if r.random() < 0.3:                           # not N(0, 1): scaled and shifted rows
        sc, sh = r.choice([0.1, 3.0]), r.choice([0., 0.7])
        xs = [(x.float() * sc + sh).to(dtype) for x in xs]
    train = r.random() < 0.85
    grad = train and r.random() < 0.5
    param_grad = grad and (learnable or "codebook_dim" in kw or kw.get("orthogonal_reg_weight", 0) > 0 or kw.get("affine_param", False))
    return VectorQuantize, kw, xs, dict(train=train, fwd_kwargs=fwd or None, grad=grad, param_grad=param_grad,
                                        unit_codebook=not kw.get("kmeans_init", False), deterministic_sampling=det)


def draw_rvq(r: random.Random, seed: int):
    grouped = r.random() < 0.25
    dim = r.choice([32, 64, 128])
    kw = dict(dim=dim, num_quantizers=r.choice([2, 3, 5]), codebook_size=r.choice([32, 64, 128]))
    if grouped:
        kw["groups"] = 2
    if r.random() < 0.4:
        kw["shared_codebook"] = True
    if r.random() < 0.25:
        kw["use_cosine_sim"] = True
    if r.random() < 0.4:
        kw["rotation_trick"] = False
    if r.random() < 0.3:
        kw["commitment_weight"] = 0.25
    if r.random() < 0.3:
        kw["decay"] = 0.95
    det = False
    if r.random() < 0.25:
        kw["threshold_ema_dead_code"] = 2
        det = True
    if not grouped and r.random() < 0.2:
        kw["codebook_dim"] = 16
    fwd = {}
    if not grouped and r.random() < 0.25:
        kw.update(quantize_dropout=True, quantize_dropout_cutoff_index=1)
        fwd["rand_quantize_dropout_fixed_seed"] = r.randint(0, 9)
    b, n = r.choice([1, 2]), r.choice([50, 120, 200])
    if det:                                     # (see draw_vq: keep the replaced codes few and distinct)
        kw["codebook_size"] = min(kw["codebook_size"], 64)
        n = max(n, -(-12 * kw["codebook_size"] // b))
    steps = r.choice([1, 2])
    bf16 = r.random() < 0.15 and not (kw.get("use_cosine_sim") and det)
    dtype = torch.bfloat16 if bf16 else torch.float32
    if not grouped and r.random() < 0.15:
        kw["accept_image_fmap"] = True
        xs = [randn(b, dim, 6, 6, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
    else:
        xs = [randn(b, n, dim, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
        if r.random() < 0.3:
            fwd["mask"] = [[i < r.randint(1, n) for i in range(n)] for _ in range(b)]
    train = r.random() < 0.85
    grad = train and r.random() < 0.5
    return (GroupedResidualVQ if grouped else ResidualVQ), kw, xs, dict(
        train=train, fwd_kwargs=fwd or None, grad=grad, param_grad=grad and "codebook_dim" in kw, unit_codebook=True, deterministic_sampling=det)


# ---- second generation (seeds >= 1000): the options as INDEPENDENT coin flips (affine_param next to the score-row losses, learnable
# codebooks next to them), more constructor switches and forward kwargs, the caller modules -------------------------------------------------
def draw_vq2(r: random.Random, seed: int):
    heads = r.choice([1, 1, 1, 2, 4])
    dim = r.choice([16, 32, 40, 64, 96, 128])
    kw = dict(dim=dim, codebook_size=r.choice([17, 32, 64, 100, 128]))
    if heads > 1:
        kw.update(heads=heads, codebook_dim=r.choice([8, 16, dim, dim]))    # (codebook_dim == dim: the masked loss has a shape)
        if r.random() < 0.5:
            kw["separate_codebook_per_head"] = True
    elif r.random() < 0.3:
        kw["codebook_dim"] = r.choice([8, 16, 24])
        if r.random() < 0.3:
            kw["layernorm_after_project_in"] = True
    cosine = r.random() < 0.25
    learnable = (not cosine) and r.random() < 0.2
    if cosine:
        kw["use_cosine_sim"] = True
    if learnable:
        kw.update(learnable_codebook=True, ema_update=False)
        if r.random() < 0.4:
            kw.update(sync_update_v=r.choice([0.1, 0.5]), rotation_trick=False)
    if "rotation_trick" not in kw and r.random() < 0.4:
        kw["rotation_trick"] = False
    if r.random() < 0.5:
        kw["commitment_weight"] = r.choice([0.25, 2.0, 0.])
    if r.random() < 0.3:
        kw["decay"] = r.choice([0.5, 0.99])
    if r.random() < 0.15:
        kw["eps"] = 1e-3
    det = False
    if not learnable and r.random() < 0.2:
        kw["threshold_ema_dead_code"] = 2
        det = True
    if not learnable and r.random() < 0.12:
        kw.update(kmeans_init=True, kmeans_iters=3)
        det = True
    if (not cosine) and r.random() < 0.15:
        kw["affine_param"] = True
    if r.random() < 0.15:
        kw["commitment_use_cross_entropy_loss"] = True
    if r.random() < 0.15:
        kw.update(codebook_diversity_loss_weight=0.5, codebook_diversity_temperature=r.choice([10., 100.]))
    if r.random() < 0.15:
        kw.update(orthogonal_reg_weight=5., orthogonal_reg_active_codes_only=(r.random() < 0.5 and not kw.get("separate_codebook_per_head", False)))
    if r.random() < 0.05:
        kw.update(stochastic_sample_codes=True, sample_codebook_temp=0.)
    if r.random() < 0.1:
        kw["route_gradients_to_input"] = False
    if r.random() < 0.05 and not learnable:
        kw["ema_update"] = False
    if r.random() < 0.05:
        kw["freeze_codebook"] = True
    b, n = r.choice([1, 2, 3]), r.choice([33, 64, 100, 150])
    if det:
        kw["codebook_size"] = min(kw["codebook_size"], 64)
        n = max(n, -(-12 * kw["codebook_size"] // b))
    steps = r.choice([1, 2])
    bf16 = r.random() < 0.12 and not (cosine and det)
    dtype = torch.bfloat16 if bf16 else torch.float32
    fwd = {}
    layout = r.random()
    if layout < 0.08:
        side = r.choice([5, 8])
        kw["accept_image_fmap"] = True
        xs = [randn(b, dim, side, side, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
    elif layout < 0.14:
        kw["accept_3d_fmap"] = True
        xs = [randn(b, dim, 3, 4, 5, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
    elif layout < 0.24:
        kw["channel_last"] = False
        xs = [randn(b, dim, n, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
    elif layout < 0.29 and not det:
        xs = [randn(b * 7, dim, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]         # one token per batch entry (vqp.py:1123-1127)
    else:
        xs = [randn(b, n, dim, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
        m = r.random()
        if m < 0.2:
            fwd["lens"] = [r.randint(1, n) for _ in range(b)]
        elif m < 0.35:
            fwd["mask"] = [[r.random() < 0.7 for _ in range(n)] for _ in range(b)]
        if fwd and r.random() < 0.3:
            kw["return_zeros_for_masked_padding"] = False
        k = r.random()
        if k < 0.07 and heads == 1:
            fwd["topk"] = r.choice([2, 3])
        elif k < 0.14:
            g = torch.Generator().manual_seed(seed)
            shape = (b, n) if heads == 1 else (b, n, heads)
            fwd["indices"] = torch.randint(0, kw["codebook_size"], shape, generator=g).tolist()
    if r.random() < 0.1:
        fwd["freeze_codebook"] = True
    if r.random() < 0.3:
        sc, sh = r.choice([0.1, 3.0]), r.choice([0., 0.7])
        xs = [(x.float() * sc + sh).to(dtype) for x in xs]
    train = r.random() < 0.85
    if kw.get("affine_param"):
        # the codebook's moment buffers are torch.empty until the first TRAINING forward fills them (vqp.py:445-448, 501-503): a fresh
        # module in eval mode searches a codebook mapped through uninitialised memory (seeds 1287, 1351)
        train = True
    grad = train and r.random() < 0.5
    param_grad = grad and (learnable or "codebook_dim" in kw or kw.get("orthogonal_reg_weight", 0) > 0 or kw.get("affine_param", False))
    return VectorQuantize, kw, xs, dict(train=train, fwd_kwargs=fwd or None, grad=grad, param_grad=param_grad,
                                        unit_codebook=not kw.get("kmeans_init", False), deterministic_sampling=det)


def draw_rvq2(r: random.Random, seed: int):
    grouped = r.random() < 0.25
    dim = r.choice([32, 64, 128])
    Q = r.choice([2, 3, 4, 6])
    kw = dict(dim=dim, num_quantizers=Q, codebook_size=r.choice([32, 64, 128]))
    if r.random() < 0.1:
        kw["codebook_size"] = tuple(r.choice([16, 32, 64]) for _ in range(Q))
        del kw["num_quantizers"]
    if grouped:
        kw["groups"] = 2
    if r.random() < 0.4 and "num_quantizers" in kw:
        kw["shared_codebook"] = True
    if r.random() < 0.25:
        kw["use_cosine_sim"] = True
    if r.random() < 0.4:
        kw["rotation_trick"] = False
    if r.random() < 0.4:
        kw["commitment_weight"] = r.choice([0.25, 0.])
    if r.random() < 0.3:
        kw["decay"] = 0.95
    det = False
    if r.random() < 0.25 and not kw.get("shared_codebook"):
        kw["threshold_ema_dead_code"] = 2
        det = True
    r.random()      # (was: k-means init inside a residual loop.  A row that ends up alone in its cluster has a ~1e-4 residual; such rows
    #  among the next layer's seeds put several codes inside the clamp(min = 1e-8) of cdist, and the reference's own indices then depend
    #  on the host's BLAS -- seeds 1054 / 1064 / 1174 / 1178: the SAME reference code gives other indices on the GPU box's host than here)
    if not kw.get("use_cosine_sim") and not det and r.random() < 0.12:
        kw.update(learnable_codebook=True, ema_update=False)
    if r.random() < 0.1:
        kw["commitment_use_cross_entropy_loss"] = True
    if r.random() < 0.1:
        kw["route_gradients_to_input"] = False
    if not grouped and r.random() < 0.2:
        kw["codebook_dim"] = 16
    fwd = {}
    if not grouped and r.random() < 0.3 and "num_quantizers" in kw:
        kw.update(quantize_dropout=True, quantize_dropout_cutoff_index=r.choice([0, 1]))
        if Q >= 4 and r.random() < 0.5:
            kw["quantize_dropout_multiple_of"] = 2
        fwd["rand_quantize_dropout_fixed_seed"] = r.randint(0, 9)
    b, n = r.choice([1, 2]), r.choice([50, 120, 200])
    if det:
        sizes = kw["codebook_size"]
        top = max(sizes) if isinstance(sizes, tuple) else sizes
        if not isinstance(sizes, tuple):
            kw["codebook_size"] = top = min(top, 64)
        n = max(n, -(-12 * top // b))
    steps = r.choice([1, 2])
    bf16 = r.random() < 0.15 and not (kw.get("use_cosine_sim") and det)
    dtype = torch.bfloat16 if bf16 else torch.float32
    xs = [randn(b, n, dim, seed=seed * 10 + s, dtype=dtype) for s in range(steps)]
    if r.random() < 0.3:
        fwd["mask"] = [[i < r.randint(1, n) for i in range(n)] for _ in range(b)]
    if r.random() < 0.1:
        fwd["freeze_codebook"] = True
    if r.random() < 0.15:
        fwd["return_all_codes"] = True
    train = r.random() < 0.85
    grad = train and r.random() < 0.5
    return (GroupedResidualVQ if grouped else ResidualVQ), kw, xs, dict(
        train=train, fwd_kwargs=fwd or None, grad=grad, param_grad=grad and ("codebook_dim" in kw or kw.get("learnable_codebook", False)),
        unit_codebook=not kw.get("kmeans_init", False), deterministic_sampling=det)


def draw_caller(r: random.Random, seed: int):
    which = r.choice(["simvq", "simvq", "rsimvq", "rpq", "hvq"])
    bf16 = False
    if which == "simvq":
        dim = r.choice([16, 32, 64])
        kw = dict(dim=dim, codebook_size=r.choice([32, 64, 200]), rotation_trick=r.random() < 0.6)
        if r.random() < 0.4:
            kw["input_to_quantize_commit_loss_weight"] = r.choice([0., 0.5])
        if r.random() < 0.3:
            kw["commitment_weight"] = 0.3
        if r.random() < 0.3:
            kw["frozen_codebook_dim"] = r.choice([8, 48])
        if r.random() < 0.3:
            kw["channel_first"] = True
            xs = [randn(2, dim, 7, 5, seed=seed * 10)]
        else:
            xs = [randn(2, r.choice([40, 130]), dim, seed=seed * 10)]
        return M.SimVQ, kw, xs, dict(grad=True, param_grad=True, train=r.random() < 0.85)
    if which == "rsimvq":
        dim = r.choice([32, 64])
        kw = dict(dim=dim, num_quantizers=r.choice([2, 4]), codebook_size=r.choice([32, 100]), rotation_trick=r.random() < 0.6)
        fwd = {}
        r.random()      # (was: quantize_dropout -- ResidualSimVQ draws the dropout index from the global RNG, no seed argument)
        if r.random() < 0.3:
            kw["channel_first"] = True
            xs = [randn(2, dim, 6, 6, seed=seed * 10)]
        else:
            xs = [randn(2, 90, dim, seed=seed * 10)]
        return M.ResidualSimVQ, kw, xs, dict(grad=True, param_grad=True, fwd_kwargs=fwd or None)
    if which == "rpq":
        dim = r.choice([32, 64, 80])
        kw = dict(dim=dim, codebook_size=r.choice([16, 32, 100]), codebook_dim=r.choice([8, 16, 32]), num_codebooks=r.choice([1, 2, 4, 8]))
        if r.random() < 0.3:
            kw["norm"] = False
        return M.RandomProjectionQuantizer, kw, [randn(2, r.choice([50, 150]), dim, seed=seed * 10)], dict()
    dim = r.choice([16, 32])
    kw = dict(dim=dim, codebook_size=r.choice([32, 64]), scales=r.choice([(1, 2, 4), (2, 4, 8), (1, 3, 6)]), accept_image_fmap=True,
              kmeans_init=r.random() < 0.5, rotation_trick=r.random() < 0.5)
    if r.random() < 0.5:
        kw["threshold_ema_dead_code"] = 0
    if r.random() < 0.4:
        kw["share_quant_resi"] = r.choice([1, 2])
    if r.random() < 0.3:
        kw["quant_resi"] = 0.
    side = max(kw["scales"])
    grad = kw["rotation_trick"] and r.random() < 0.7
    return M.HierarchicalVQ, kw, [randn(2, dim, side, side, seed=seed * 10 + s) for s in range(r.choice([1, 2]))], dict(
        grad=grad, deterministic_sampling=True, unit_codebook=not kw["kmeans_init"])


if __name__ == "__main__":
    torch.set_num_threads(8)
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    M.ONLY.clear()
    made = skipped = 0
    for seed in range(first, first + count):
        r = random.Random(9000 + seed)
        if seed >= 1000:
            cls, kw, xs, opts = (draw_vq2, draw_vq2, draw_rvq2, draw_vq2, draw_rvq2, draw_caller)[seed % 6](r, 500 + seed)
            opts.setdefault("fwd_kwargs", None)
        else:
            cls, kw, xs, opts = (draw_rvq if seed % 3 == 2 else draw_vq)(r, 500 + seed)
        name = f"combo_{seed:03d}"
        why = None
        if kw.get("shared_codebook") and kw.get("threshold_ema_dead_code"):
            # every stage replaces its dead codes with rows of ITS input inside the loop (vqp.py:641): the shared codebook collects
            # rows of successive residuals, pairs of codes 5e-7 apart appear, and dozens of rows per stage are decided by the last bit of
            # the BLAS call (seen on seeds 170, 206: the reference's indices are the argmin of its own sgemm, ours of the fp32 chain)
            why = "shared codebook + dead-code replacement: near-duplicate codes, the search is a coin toss between them"
        if kw.get("kmeans_init") and kw.get("learnable_codebook"):
            # the k-means means differ in the last bit with the summation order, and the codes' gradient 2 (q - x) / n is a difference of
            # nearly equal numbers right after the initialisation (seeds 4, 94: 6e-5 .. 3e-3 of the gradient's scale)
            why = "k-means init + learnable codebook: the gradient right after the init amplifies the last bit of the means"
        if why:
            skipped += 1
            print(f"{name}: SKIPPED, {why}")
            continue
        try:
            M.run_case(name, cls, kw, xs, **opts)
            made += 1
            print("    ", cls.__name__, kw, {k: v for k, v in opts.items() if v and k != "fwd_kwargs"}, sorted((opts.get("fwd_kwargs") or {}).keys()))
        except Exception as e:                                    # the reference rejects the combination
            skipped += 1
            print(f"{name}: SKIPPED, the reference raises {type(e).__name__}: {str(e)[:120]}   {cls.__name__} {kw}")
            p = os.path.join(HERE, name + ".npz")
            if os.path.exists(p):
                os.remove(p)
    print(f"{made} fixtures, {skipped} draws rejected by the reference")
