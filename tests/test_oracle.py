"""CPU: pins the oracle (oracle/vq_oracle.{c,py}) to the golden vectors produced by the live reference.

 * mode="aten"  (the reference's own ATen/MKL op sequence restated) must reproduce every fixture:
   indices identical, floats to 1e-6 (bit-identical on the host that generated them).
 * mode="chain" (the deterministic C restatement the HIP kernels are compared with bit-for-bit) must
   give identical indices on every fixture and floats within the north-star tolerance (1e-5).
"""
import pytest
import torch

import golden_util as G
from oracle import vq_oracle as O


def _close(a, b, tol, what):
    a, b = a.double(), b.double()
    scale = max(b.abs().max().item(), 1e-12)
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


# layout / projection / multi-head fixtures are module-level only (checked on the GPU against the fixture itself)
_MODULE_ONLY = ("vq_fmap", "vq_proj", "vq_heads", "vq_heads_sep", "vq_3d", "vq_channel_first",
                "vq_learnable", "vq_learnable_sync_v", "vq_orthogonal", "vq_inplace_opt", "vq_bridge",
                "simvq", "simvq_ste_channel_first", "residual_simvq", "rpq", "hvq", "hvq_nokmeans",
                "vq_ce_commit", "vq_diversity", "vq_topk", "vq_topk_cos", "vq_indices_ce", "vq_stochastic_temp0", "vq_gumbel_st", "rvq_beam", "rvq_beam_shared_mask", "vq_affine",
                "vq_heads_ce", "vq_heads_diversity", "vq_heads_gumbel_st", "vq_heads_affine", "vq_heads_sep_ce", "vq_heads_sep_diversity",
                "vq_heads_sep_learnable", "vq_heads_sep_affine",
                "vq_heads_sep_ce_kmeans_lens", "vq_affine_learnable", "vq_affine_ce", "vq_affine_diversity", "vq_affine_topk",
                "vq_affine_learnable_ce_lens", "vq_heads_sep_mask_origdim", "vq_heads_mask_origdim",
                "rvq_qinco", "rvq_qinco_eval", "rvq_grad_mask", "vq_cos_transform_nograd", "rvq_dropout", "rpq_indices", "vq_ce_kmeans")


@pytest.mark.parametrize("name", [n for n in G.names() if n not in _MODULE_ONLY and not n.startswith("combo_")])
@pytest.mark.parametrize("mode", ["aten", "chain"])
def test_oracle_reproduces_reference(name, mode):
    fx = G.Fixture(name)
    outs, final = G.run_oracle(fx, mode)
    tol = 1e-6 if mode == "aten" else 1e-5
    if fx.bf16:
        tol = 1e-2
    if fx.meta["kwargs"].get("dim", 0) > 512 and fx.meta["grad"]:
        tol = max(tol, 1e-5)        # (the rotation trick's row reductions over 768+ elements: torch's own result moves by > 1e-6 with the order)
    for s, o in enumerate(outs):
        want_idx = fx.t(f"idx{s}")
        assert torch.equal(o["idx"], want_idx), f"step {s}: {(o['idx'] != want_idx).sum().item()} index mismatches"
        _close(o["loss"].reshape(-1), fx.t(f"loss{s}").reshape(-1), tol, f"loss step {s}")
        if fx.has(f"q{s}"):
            _close(o["q"].float(), fx.t(f"q{s}").float(), tol, f"quantized step {s}")
        else:
            assert abs(o["q"].double().sum().item() - float(fx.arr[f"qsum{s}"])) <= 1e-6 * max(1.0, abs(float(fx.arr[f"qsum{s}"])))
        if "gx" in o:
            _close(o["gx"].float(), fx.t(f"gx{s}").float(), tol if mode == "chain" else 1e-5, f"grad_x step {s}")
    if fx.meta["train"]:
        after = fx.state("after")
        for k, v in final.items():
            if k.endswith("initted"):
                continue
            _close(v.float(), after[k].float(), tol, k)


@pytest.mark.parametrize("cosine", [False, True])
def test_oracle_onehot_quantize_mode_is_the_gather_bit_for_bit(cosine):
    """quantize_mode="onehot" restates the reference's TRAINING branch literally -- F.one_hot(ind).type(dtype) contracted with the
    codebook (vqp.py:142, 766), the same tensor feeding the EMA einsum (:602-606): the op sequence bench.py's cpu_baseline times
    (the gather form runs 2 of the reference's 3 N*C*D contractions).  Outputs and state equal the gather form's."""
    torch.manual_seed(0)
    cfg = O.VQConfig(dim=32, codebook_size=64, use_cosine_sim=cosine)
    e = torch.randn(1, 64, 32)
    if cosine:
        e = torch.nn.functional.normalize(e, dim=-1)
    x = torch.randn(2, 300, 32)
    outs = []
    for qm in ("gather", "onehot"):
        st = O.VQState(embed=e.clone(), embed_avg=e.clone(), cluster_size=torch.ones(1, 64))
        with torch.no_grad():
            q, ind, loss = O.vq_forward(st, cfg, x, quantize_mode=qm)
        outs.append((q, ind, loss, st.embed.clone(), st.embed_avg.clone(), st.cluster_size.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D", [2, 8, 20, 32, 40, 64, 100, 128, 256, 384, 512, 520, 640, 768, 1000, 1024, 1536, 2048])
def test_c_sumsq_is_aten_order(D):
    x = torch.randn(2000, D, generator=torch.Generator().manual_seed(D))
    assert torch.equal(O.c_row_sumsq(x), (x ** 2).sum(-1))


def test_chain_vs_aten_index_agreement_audit():
    """default (tiny, tie-prone) codebook, N = 16384: the FMA-chain dot product may only disagree with
    MKL's on rows whose two best scores are within a couple of ulps -- classify every mismatch."""
    g = torch.Generator().manual_seed(0)
    N, C, D = 16384, 1024, 256
    x = torch.randn(N, D, generator=g)
    e = (torch.rand(C, D, generator=g) * 2 - 1) * (6.0 / (C * D)) ** 0.5
    ia = O.neg_cdist(x[None], e[None]).argmax(-1)[0]
    ic, _ = O.c_assign(x, e)
    audit = G.classify_mismatches(x, e, ia, ic)
    assert len(audit) <= N * 1e-3
    assert all(gap <= 2.0 for *_, gap in audit), audit


def test_oracle_first_occurrence_on_ties():
    x = torch.randn(100, 32, generator=torch.Generator().manual_seed(3))
    e = torch.randn(40, 32, generator=torch.Generator().manual_seed(4))
    e2 = torch.cat([e, e])
    assert int(O.c_assign(x, e2)[0].max()) < 40
    assert torch.equal(O.c_assign(x, e2)[0], O.neg_cdist(x[None], e2[None]).argmax(-1)[0])


@pytest.mark.parametrize("name", ["rvq_big_rot"])
def test_oracle_aten_reproduces_the_big_residual_golden(name):
    """The size where near-ties show up (65 536 rows x 8 stages x 1024 shared codes, default init; tests/golden/big, made by the
    live reference with an input that requires grad + rotation trick): the oracle's `aten` mode issues the reference's own op
    sequence, so on the host that made the fixture it reproduces every index (and anywhere else all but near-ties of MKL's
    blocked dot products -- bounded here, audited on the GPU side by test_gpu_modules)."""
    import hashlib, json, os
    import numpy as np
    z = np.load(os.path.join(G.GOLDEN, "big", name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    x = torch.randn(*meta["shape"], generator=torch.Generator().manual_seed(meta["seed"]))
    if hashlib.sha1(x.numpy().tobytes()).hexdigest() != meta["xsha"]:
        pytest.skip("torch's CPU generator draws other numbers here than in the container that made the fixture")
    kw = meta["kwargs"]
    sd = {k[len("before/"):]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("before/")}
    sd["layers.0._codebook.embed_avg"] = sd["layers.0._codebook.embed"].clone()
    st = O.VQState.from_state_dict(sd, "layers.0._codebook.")
    cfg = G.oracle_cfg(kw)
    xin = x.clone().requires_grad_(meta["grad"])
    _, idx, losses = O.rvq_forward([st] * kw["num_quantizers"], cfg, xin, shared_codebook=True, training=True,
                                   assign_mode="aten", stats_mode="aten")
    want = torch.from_numpy(z["idx"].astype(np.int64))
    rows = int((idx != want).any(-1).sum())
    assert rows <= 96, f"{rows} of {idx.numel() // idx.shape[-1]} rows continue with another code than the reference's"
    _close(losses.detach(), torch.from_numpy(np.array(z["losses"])), 1e-5, "losses")


# ---- bf16 rows under the rotation trick: the sequence of roundings the routing kernels apply ----------------------------------------
@pytest.mark.parametrize("D", [16, 64, 128, 256])
@pytest.mark.parametrize("scale", [1.0, 0.05, 30.0])
def test_bf16_rotation_restated_op_by_op_equals_torchs_bf16_tensor_ops(D, scale):
    """The reference's rotate_to runs on bf16 TENSORS when the rows are bf16 (vqp.py:1178, 287-318): every op rounds.  The oracle's
    restatement with explicit roundings (what csrc/vq_route_math.h implements) gives the same bits as torch's bf16 ops, and the fp32
    formula does not (~2 % of the output's magnitude): the reason the kernels round op by op."""
    g = torch.Generator().manual_seed(D)
    e = (torch.randn(2048, D, generator=g) * scale).bfloat16()
    q = (torch.randn(2048, D, generator=g) * scale * 0.7).bfloat16()
    ref = O.rotate_to(e, q).float()                               # torch's bf16 ops == the reference's graph
    mine = O.rotate_to_bf16_ops(e.float(), q.float())
    rows_off = (ref != mine).any(-1).sum().item()
    assert rows_off <= 4, f"{rows_off} of 2048 rows differ"       # (only where the order of a row's fp32 sum moves a rounding)
    plain = O.rotate_to(e.float(), q.float())
    assert ((plain - ref).abs().max() / ref.abs().max()).item() > 5e-3


def test_a_quotient_of_two_bf16_values_is_never_near_a_bf16_rounding_midpoint():
    """Basis of the kernels' `a * rcp(b)` in place of an IEEE division (vq_rot_frame<..., BF16>): over all 128 x 128 pairs of 8-bit
    significands the quotient stays >= 2^-17 (relative) away from the midpoint of two neighbouring bf16 values, far more than the
    < 2^-22 error of v_rcp_f32 followed by one multiply -- so bf16(a * rcp(b)) == bf16(fp32(a / b)) bit for bit."""
    from fractions import Fraction
    best = Fraction(1)
    for a in range(128, 256):
        for b in range(128, 256):
            x = Fraction(a, b)
            while x < 1:
                x *= 2
            x *= 128                                              # bf16 values of this binade are the integers 128 .. 255 now
            j = int(x)
            d = abs(x - (Fraction(2 * j + 1, 2)))                 # distance to the midpoint between j and j + 1
            if x != j:
                best = min(best, d / x)
    assert best > Fraction(1, 1 << 17), float(best)
