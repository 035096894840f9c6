"""CPU (no GPU): the C-ABI library loads and exports every symbol include/vqhip.h declares, argument
validation happens before any HIP call, the host-side mirror keeps the reference's constructor / state_dict
contract, and the product refuses to run without the GPU."""
import ctypes
import os
import re

import pytest
import torch

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "vqhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vqhip_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vector_quantize_pytorch_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/vqhip.h but not exported"
    assert set(names) == set(_lib.EXPORTS)
    assert b"gfx950" in L.vqhip_version()


def test_host_only_entry_points():
    from vector_quantize_pytorch_amd import _lib
    L = _lib.lib()
    tile = 128 * 256 + 1024   # fp32 A-operand tile == bf16 hi/lo screening tile, bytes (csrc/vqhip_internal.h)
    tile16 = 64 * 256 + 1024  # fp16 single-pass screening tile
    assert L.vqhip_packed_bytes(1024, 256) == (32 * tile + 4096) + (1024 * 256 * 2 + 8192) + 64 + (32 * tile16 + 8192)
    assert L.vqhip_screen_supported(1 << 20, 256, 1024) == 1 and L.vqhip_screen_supported(1 << 20, 512, 1024) == 1 and L.vqhip_screen_supported(1 << 20, 96, 1024) == 0
    assert L.vqhip_screen_partials(1 << 20, 1) == (1 << 20) // 256 + 512   # bf16: screen workgroups + finish workgroups of the exact pass
    assert L.vqhip_screen_partials(1 << 20, 0) == (1 << 20) // 256 + 512   # fp32: 8 waves x 32 rows per workgroup
    assert L.vqhip_screen_workspace_bytes(1000) >= 16 + 4 * 1000 + 8 * 1000
    assert L.vqhip_packed_bytes(33, 100) == (2 * (128 * 128 + 1024) + 4096 + (33 * 100 * 2 + 8) + 8192 + 64
                                             + 8 * (64 * 128 + 1024) + 8192)   # D padded to 128, C to 64 (fp16 tiles: to 8 tiles); bf16 copy 16-byte rounded
    assert L.vqhip_packed_bytes(16, 2049) == 0                             # unsupported D
    # wide dims (csrc/vq_wide.hip, round 5): ||c||^2 [C] floats (256-byte padded) + the bf16 copy [C, D] (256-byte padded) + 256
    assert L.vqhip_packed_bytes(16, 513) == 256 + (16 * 513 * 2 + 255) // 256 * 256 + 256
    assert L.vqhip_screen_supported(1 << 20, 1024, 1024) == 0 and L.vqhip_vq_step_supported(0, 1 << 20, 1024, 1024) == 0
    assert L.vqhip_assign_blocks(0) == 0 and L.vqhip_assign_blocks(1) == 1 and L.vqhip_assign_blocks(129) == 2


def test_argument_validation_precedes_any_gpu_work():
    from vector_quantize_pytorch_amd import _lib
    L = _lib.lib()
    null = ctypes.c_void_p(0)
    rc = L.vqhip_assign(null, 0, 16, 64, 64, null, null, 8, 0, null, null, 0, 64, null, null, null, null, null)
    assert rc == -1 and b"null" in L.vqhip_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.vqhip_assign(p, 0, 16, 4096, 4096, p, p, 8, 0, p, null, 0, 64, null, null, null, null, null) == -2   # D too large (> 2048)
    assert L.vqhip_assign(p, 7, 16, 64, 64, p, p, 8, 0, p, null, 0, 64, null, null, null, null, null) == -1      # dtype
    assert L.vqhip_assign(p, 0, 0, 64, 64, p, p, 8, 0, p, null, 0, 64, null, null, null, null, null) == 0        # N == 0: no-op
    assert L.vqhip_pack_codebook(null, 8, 64, null, null) == -1
    assert L.vqhip_ema_finalize(p, p, p, null, null, null, 8, 64, 0.2, 1e-5, 0, 1, 1, null, null) == -1          # do_lerp w/o stats
    assert L.vqhip_decode_sum(null, 4, 1, null, 0, 8, 64, null, 0, 64, null) == -1


@pytest.mark.parametrize("name", G.names())
def test_state_dict_contract_matches_reference(name):
    """every fixture of the live reference -- the named cases and the random option combinations: the constructor takes the reference's
    kwargs (nothing raises NotImplementedError on the host side) and the module has the reference's state_dict keys, shapes, dtypes"""
    import vector_quantize_pytorch_amd as A
    fx = G.Fixture(name)
    mod = G.build_special(name, A) if fx.meta.get("build") else getattr(A, fx.meta["cls"])(**fx.kwargs)
    ref = fx.state("before")
    mine = mod.state_dict()
    assert list(mine.keys()) == list(ref.keys()) or set(mine.keys()) == set(ref.keys())
    for k in ref:
        assert mine[k].shape == ref[k].shape and mine[k].dtype == ref[k].dtype, k
    mod.load_state_dict(ref, strict=True)


def test_default_init_matches_reference_statistics():
    from vector_quantize_pytorch_amd import VectorQuantize
    vq = VectorQuantize(dim=256, codebook_size=512)
    e = vq._codebook.embed
    assert e.shape == (1, 512, 256) and e.abs().max().item() <= (6.0 / (512 * 256)) ** 0.5 + 1e-7   # kaiming-uniform bound (vqp.py:112-115)
    assert torch.equal(vq._codebook.cluster_size, torch.ones(1, 512)) and bool(vq._codebook.initted)
    vc = VectorQuantize(dim=64, codebook_size=32, use_cosine_sim=True)
    assert torch.allclose(vc._codebook.embed.norm(dim=-1), torch.ones(1, 32), atol=1e-6)
    vk = VectorQuantize(dim=64, codebook_size=32, kmeans_init=True)
    assert not bool(vk._codebook.initted) and vk._codebook.embed.abs().sum().item() == 0
    # VectorQuantize default threshold is 0 while Codebook's is 2 (vqp.py:818 vs :360)
    assert vq._codebook.threshold_ema_dead_code == 0


def test_unsupported_options_fail_loudly_and_cpu_is_refused():
    from vector_quantize_pytorch_amd import ResidualVQ, VectorQuantize
    from vector_quantize_pytorch_amd._lib import VQHipError
    # (round 5: several heads with affine_param / the score-row options / learnable codebooks are built -- goldens vq_heads_*; what is
    #  left raising: forward(topk=) with heads > 1, which fails in the reference itself, checked on the GPU)
    for kw in (dict(affine_param=True, heads=2, codebook_dim=16), dict(stochastic_sample_codes=True, heads=2, codebook_dim=16),
               dict(commitment_use_cross_entropy_loss=True, heads=2, codebook_dim=16, separate_codebook_per_head=True),
               dict(learnable_codebook=True, ema_update=False, heads=2, codebook_dim=16, separate_codebook_per_head=True)):
        VectorQuantize(dim=32, codebook_size=16, **kw)
    with pytest.raises(NotImplementedError):
        ResidualVQ(dim=32, num_quantizers=2, codebook_size=16, implicit_neural_codebook=True, beam_size=2)
    qinco = ResidualVQ(dim=32, num_quantizers=3, codebook_size=16, implicit_neural_codebook=True, mlp_kwargs=dict(depth=1))
    assert len(qinco.mlps) == 2 and "mlps.1.proj_in.weight" in qinco.state_dict() and qinco.layers[0].learnable_codebook   # rvq.py:207-211, 288-289
    with pytest.raises(AssertionError):
        ResidualVQ(dim=32, num_quantizers=2, codebook_size=16, heads=2)
    for kw in (dict(learnable_codebook=True), dict(learnable_codebook=True, ema_update=False, use_cosine_sim=True),
               dict(sync_update_v=0.5), dict(directional_reparam=True)):       # the reference's own cross-flag asserts
        with pytest.raises(AssertionError):
            VectorQuantize(dim=32, codebook_size=16, **kw)
    vq = VectorQuantize(dim=32, codebook_size=16, learnable_codebook=True, ema_update=False)
    assert isinstance(vq._codebook.embed, torch.nn.Parameter) and "_codebook.embed" in vq.state_dict()
    vq = VectorQuantize(dim=32, codebook_size=16)
    with pytest.raises(VQHipError, match="no CPU fallback"):
        vq(torch.randn(1, 4, 32))
    with pytest.raises(NotImplementedError):                    # a per-row codebook together with an option that reads the score row
        vq(torch.randn(1, 4, 32), codebook_transform_fn=lambda e: e, topk=2)
    with pytest.raises(VQHipError, match="no CPU fallback"):
        vq(torch.randn(1, 4, 32), codebook_transform_fn=lambda e: e[:, None, None].expand(1, 1, 4, 16, 32))


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/ (comments may cite it)"""
    pkg = os.path.join(ROOT, "vector_quantize_pytorch_amd")
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle|#\s*include.*oracle)|libvqoracle|CDLL\(.*oracle", re.M)
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                assert not bad.search(open(os.path.join(dp, f)).read()), f


def test_as_rows_views():
    from vector_quantize_pytorch_amd._lib import as_rows
    x = torch.randn(4, 10, 64)
    t, N, D, ld = as_rows(x)
    assert (N, D, ld) == (40, 64, 64) and t.data_ptr() == x.data_ptr()
    c = x.chunk(4, -1)[1]
    t, N, D, ld = as_rows(c)
    assert (N, D, ld) == (40, 16, 64) and t.data_ptr() == c.data_ptr()         # no copy for feature chunks
    tr = x.transpose(1, 2)
    t, N, D, ld = as_rows(tr)
    assert (N, D, ld) == (4 * 64, 10, 10) and t.is_contiguous()                  # copied
    t, N, D, ld = as_rows(x[:, ::2])
    assert (N, ld) == (20, 128) and t.data_ptr() == x.data_ptr()                 # uniform row stride 128: still no copy
    t, N, D, ld = as_rows(x[:, :3])
    assert (N, ld) == (12, 64) and t.is_contiguous() and t.data_ptr() != x.data_ptr()   # ragged leading stride -> copy


def test_key_packing_orders_like_argmax():
    from vector_quantize_pytorch_amd.parallel import pack_score_index, unpack_score_index
    g = torch.Generator().manual_seed(0)
    s = torch.randn(4096, generator=g)
    s[::7] = s[3]                       # force ties
    s[5] = float("-inf"); s[6] = 0.0; s[8] = -0.0
    idx = torch.randperm(4096, generator=g)
    key = pack_score_index(s, idx)
    s2, i2 = unpack_score_index(key)
    assert torch.equal(s2.view(torch.int32), s.view(torch.int32)) and torch.equal(i2, idx)
    order = torch.argsort(key, descending=True)
    ss, ii = s[order], idx[order]
    assert (ss[:-1] >= ss[1:]).all()
    same = ss[:-1] == ss[1:]
    assert (ii[:-1][same] < ii[1:][same]).all()     # among equal scores the lowest index has the largest key


@pytest.mark.skipif(not os.path.isdir("/root/reference/vector_quantize_pytorch"), reason="the live reference only exists in the build container")
def test_reference_residual_vq_forward_indices_is_broken_upstream():
    """Why ResidualVQ.forward(indices=) raises here instead of mirroring the reference: the reference's own path cannot run --
    rvq.py:493 unpacks three values from the (quantize, ce_loss) pair that VectorQuantize.forward returns for indices= (vqp.py:1261)."""
    import sys
    sys.path[:0] = [os.path.join(ROOT, "oracle", "refshim"), "/root/reference"]
    try:
        from vector_quantize_pytorch import ResidualVQ as RefRVQ
        m = RefRVQ(dim=16, num_quantizers=2, codebook_size=8)
        with pytest.raises(ValueError, match="not enough values to unpack"):
            m(torch.randn(1, 4, 16), indices=torch.zeros(1, 4, 2, dtype=torch.long))
    finally:
        del sys.path[:2]


def test_every_entry_point_has_a_declared_ctypes_signature():
    """An entry point called without `argtypes` gets its int64 / pointer arguments passed as C ints: silently wrong strides on the
    GPU box (this caught nothing on CPU before it was added)."""
    from vector_quantize_pytorch_amd import _lib
    L = _lib.lib()
    for sym in _lib.EXPORTS:
        if sym in ("vqhip_version", "vqhip_last_error"):
            continue
        assert getattr(L, sym).argtypes is not None, f"{sym}: no argtypes declared in _lib.lib()"


def test_expire_pick_permutation_is_a_bijection_of_the_rows():
    """The row vqhip_expire_pick hands code c (csrc: vq_expire_pick_kernel): the affine map t -> (a t + b) mod p of Z_p, p the smallest
    prime >= n, cycle-walked into [0, n), followed by a keyed Feistel permutation of [0, n).  Restated here in Python: for C <= n the codes take pairwise distinct rows (the reference
    samples without replacement, vqp.py:180-188), for every (a, b) in [1, p)."""
    import random
    from vector_quantize_pytorch_amd._lib import _next_prime
    assert [_next_prime(n) for n in (1, 2, 3, 4, 8, 9, 90, 1024, 8192)] == [2, 2, 3, 5, 11, 11, 97, 1031, 8209]

    M32 = 0xFFFFFFFF

    def pick(c, a, b, p, n):
        v = (a * (c % p) + b) % p
        it = 0
        while v >= n and it < 64:
            v = (a * v + b) % p
            it += 1
        v = v if v < n else v % n
        if n <= 2:
            return v
        # round 5: a 4-round Feistel network on the smallest even-width power-of-two domain >= n, cycle-walked into [0, n), on top of
        # the affine map (whose picks for neighbouring codes form an arithmetic progression) -- restated from the kernel
        kb = 1
        while (1 << kb) < n:
            kb += 1
        kb += kb & 1
        hb, mask = kb >> 1, (1 << (kb >> 1)) - 1
        w = v
        for _ in range(64):
            Lh, R = (w >> hb) & mask, w & mask
            for r in range(4):
                f = (R * 0x9E3779B1 + ((b if r & 1 else a) & M32) + 0x85EBCA6B * (r + 1) + (((a if r & 2 else b) >> 17) & M32)) & M32
                f ^= f >> 15; f = (f * 0x2C1B3C6D) & M32; f ^= f >> 12; f = (f * 0x297A2D39) & M32; f ^= f >> 15
                Lh, R = R, Lh ^ (f & mask)
            w = (Lh << hb) | R
            if w < n:
                return w
        return v

    rng = random.Random(0)
    for n in (1, 2, 7, 90, 1000, 4099):
        p = _next_prime(n)
        for _ in range(20):
            a, b = rng.randrange(1, p), rng.randrange(1, p)
            C = min(n, 512)
            rows = [pick(c, a, b, p, n) for c in range(C)]
            assert all(0 <= r < n for r in rows)
            # distinct unless the walk was cut short by the 64-step cap (a short cycle of the map inside [n, p): not for these sizes)
            assert len(set(rows)) == C, (n, p, a, b)
            if n >= 1000:       # neighbouring codes no longer take rows a fixed stride apart
                steps = {(rows[c + 1] - rows[c]) % n for c in range(C - 1)}
                assert len(steps) > C // 2, (n, a, b, len(steps))


def test_other_float_dtypes_are_computed_in_float32_and_cast_back():
    """The decorator around the modules' forward (vector_quantize.other_float_dtypes_as_fp32; reference: x.float() ... .type(dtype),
    vqp.py:690, 1178): float16 / float64 inputs reach the wrapped forward as float32, positional or as x=, the code tensors come back
    in the input's dtype, indices / losses / LossBreakdown-like extras untouched; float32 and bfloat16 pass through as they are."""
    from vector_quantize_pytorch_amd.vector_quantize import other_float_dtypes_as_fp32
    seen = []

    class M:
        @other_float_dtypes_as_fp32
        def forward(self, x, flag=False):
            seen.append(x.dtype)
            out = (x * 2, torch.zeros(x.shape[:-1], dtype=torch.long), torch.zeros(()))
            return (*out, torch.stack([x, x])) if flag else out

    m = M()
    for dt in (torch.float16, torch.float64):
        x = torch.randn(2, 3, 4).to(dt)
        q, ind, loss = m.forward(x)
        assert seen[-1] == torch.float32 and q.dtype == dt and ind.dtype == torch.long and loss.dtype == torch.float32
        q, ind, loss, codes = m.forward(x=x, flag=True)
        assert seen[-1] == torch.float32 and q.dtype == dt and codes.dtype == dt and codes.shape == (2, 2, 3, 4)
    for dt in (torch.float32, torch.bfloat16):
        q, _, _ = m.forward(torch.randn(2, 3, 4).to(dt))
        assert seen[-1] == dt and q.dtype == dt


def test_transposed_view_detection_is_for_gpu_tensors_only():
    from vector_quantize_pytorch_amd import _lib
    t = torch.randn(3, 8, 5).transpose(1, 2)
    assert not _lib.is_transposed_view(t)                  # CPU tensor: the product has no CPU path
    assert _lib.rows_contiguous(torch.randn(4, 4)).is_contiguous()


def test_committed_option_combinations_are_the_generators_draws():
    """tests/golden/combo_<seed>.npz is what `python tests/golden/make_combo.py <seed> 1` draws: the constructor kwargs stored in every
    committed fixture equal the generator's draw for that seed (build container only: the generator imports the live reference)."""
    import os
    import random
    import sys
    if not os.path.isdir("/root/reference/vector_quantize_pytorch"):
        pytest.skip("the live reference is not mounted here")
    sys.path.insert(0, os.path.join(G.GOLDEN))
    try:
        import make_combo as MC
    except Exception as e:                                   # (the reference or its einx stand-in failed to import)
        pytest.skip(f"generator not importable: {e}")
    finally:
        sys.path.pop(0)
    checked = 0
    for name in G.names():
        if not name.startswith("combo_"):
            continue
        seed = int(name[len("combo_"):])
        r = random.Random(9000 + seed)
        if seed >= 1000:
            draw = (MC.draw_vq2, MC.draw_vq2, MC.draw_rvq2, MC.draw_vq2, MC.draw_rvq2, MC.draw_caller)[seed % 6]
        else:
            draw = MC.draw_rvq if seed % 3 == 2 else MC.draw_vq
        cls, kw, xs, opts = draw(r, 500 + seed)
        fx = G.Fixture(name)
        assert fx.meta["cls"] == cls.__name__, name
        want = {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()}
        assert fx.meta["kwargs"] == want, (name, fx.meta["kwargs"], want)
        assert fx.meta["steps"] == len(xs) and fx.meta["grad"] == bool(opts.get("grad", False)), name
        checked += 1
    assert checked >= 60
