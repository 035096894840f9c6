"""Loader / drivers for the golden fixtures in tests/golden (made by tests/golden/make_golden.py from
the live reference)."""
import json
import os

import numpy as np
import torch

from oracle import vq_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.name = name
        self.bf16 = self.meta["bf16"]
        self.arr = {k: z[k] for k in z.files if k != "meta"}

    def t(self, key, like_x=False):
        a = torch.from_numpy(np.array(self.arr[key]))
        if a.dtype == torch.int16:            # raw bf16 bits
            a = a.view(torch.bfloat16)
        return a

    def has(self, key):
        return key in self.arr

    def state(self, which="before"):
        sd = {k[len(which) + 1:]: self.t(k) for k in self.arr if k.startswith(which + "/")}
        if which == "before":                  # embed_avg omitted when equal to embed
            for k in list(sd):
                if k.endswith("embed") and (k + "_avg") not in sd:
                    sd[k + "_avg"] = sd[k].clone()
        if self.meta["kwargs"].get("shared_codebook"):
            nq = self.meta["kwargs"]["num_quantizers"]
            for k in list(sd):
                if k.startswith("layers.0."):
                    for i in range(1, nq):
                        sd[f"layers.{i}." + k[len("layers.0."):]] = sd[k]
        return sd

    @property
    def kwargs(self):
        kw = dict(self.meta["kwargs"])
        if isinstance(kw.get("codebook_size"), list):
            kw["codebook_size"] = tuple(kw["codebook_size"])
        return kw

    def fwd_kwargs(self, device="cpu"):
        fk = dict(self.meta["fwd_kwargs"])
        if "lens" in fk:
            fk["lens"] = torch.tensor(fk["lens"], device=device)
        if "indices" in fk:
            fk["indices"] = torch.tensor(fk["indices"], device=device)
        if "mask" in fk:
            fk["mask"] = torch.tensor(fk["mask"], device=device)
        return fk


class TransformCaller(torch.nn.Module):
    """TEST HARNESS (used by tests/golden/make_golden.py around the LIVE reference and by the GPU tests around this repo's module):
    a VectorQuantize called with a fixed, row-dependent `codebook_transform_fn` (vqp.py:729-738) -- every row searches the codebook
    shifted by a tenth of the (detached) row itself."""

    def __init__(self, vq):
        super().__init__()
        self.vq = vq

    def forward(self, x, **kw):
        shift = 0.1 * x.detach()
        # codes [h, c, d] -> one codebook per row, [h, b, n, c, d] (the layout vqp.py:731 expects)
        return self.vq(x, codebook_transform_fn=lambda codes: codes[:, None, None] + shift[None, ..., None, :], **kw)


def build_special(name, A):
    """fixtures whose constructor takes callables / modules (not serialisable in the fixture's kwargs)"""
    if name == "vq_cos_transform_nograd":
        return TransformCaller(A.VectorQuantize(dim=32, codebook_size=64, use_cosine_sim=True))
    if name == "vq_inplace_opt":
        return A.VectorQuantize(dim=32, codebook_size=64, learnable_codebook=True, ema_update=False,
                                in_place_codebook_optimizer=lambda p: torch.optim.SGD(p, lr=0.5))
    if name == "vq_bridge":
        return A.VectorQuantize(dim=32, codebook_size=64, vq_bridge=torch.nn.Linear(32, 32))
    raise KeyError(name)


def first_rows(samples, num):
    n = samples.shape[1]
    if n >= num:
        return samples[:, :num].clone()
    reps = -(-num // n)
    return samples.repeat(1, reps, 1)[:, :num].clone()


# ---- oracle driver -------------------------------------------------------------------------------
def oracle_cfg(kw, D=None):
    return O.VQConfig(
        dim=kw["dim"] if D is None else D, codebook_size=kw["codebook_size"], use_cosine_sim=kw.get("use_cosine_sim", False),
        decay=kw.get("decay", 0.8), eps=kw.get("eps", 1e-5), threshold_ema_dead_code=kw.get("threshold_ema_dead_code", 0),
        kmeans_init=kw.get("kmeans_init", False), kmeans_iters=kw.get("kmeans_iters", 10),
        commitment_weight=kw.get("commitment_weight", 1.0), rotation_trick=kw.get("rotation_trick", kw["dim"] > 1),
        manual_ema_update=kw.get("shared_codebook", False))


def run_oracle(fx: Fixture, mode="aten"):
    """Returns list of per-step dict(q, idx, loss[, gx]) and the final state dict (same keys as the fixture)."""
    kw, meta = fx.kwargs, fx.meta
    sd = fx.state("before")
    samp = dict(sample_fn=first_rows, replace_sample_fn=first_rows) if meta["deterministic_sampling"] else {}
    stats = "aten" if mode == "aten" else "double"
    outs = []
    cls = meta["cls"]
    if cls == "VectorQuantize":
        cfg = oracle_cfg(kw)
        st = O.VQState.from_state_dict(sd)
        for s in range(meta["steps"]):
            x = fx.t(f"x{s}").clone()
            if meta["grad"]:
                x.requires_grad_(True)
            fk = fx.fwd_kwargs()
            q, idx, loss = O.vq_forward(st, cfg, x, training=meta["train"], assign_mode=mode, stats_mode=stats, **fk, **samp)
            o = dict(q=q.detach(), idx=idx, loss=loss.detach())
            if meta["grad"]:
                (loss.sum() * 3.0 + (q * fx.t(f"gw{s}")).sum()).backward()
                o["gx"] = x.grad
            outs.append(o)
        final = {"_codebook.embed": st.embed, "_codebook.embed_avg": st.embed_avg, "_codebook.cluster_size": st.cluster_size,
                 "_codebook.initted": torch.tensor(st.initted)}
        return outs, final

    def rvq(kw, sd, prefix, xs_fn, steps):
        sizes = kw["codebook_size"]
        Q = len(sizes) if isinstance(sizes, tuple) else kw["num_quantizers"]
        sizes = sizes if isinstance(sizes, tuple) else (sizes,) * Q
        shared = kw.get("shared_codebook", False)
        if shared:
            st0 = O.VQState.from_state_dict(sd, prefix + "layers.0._codebook.")
            states = [st0] * Q
        else:
            states = [O.VQState.from_state_dict(sd, prefix + f"layers.{i}._codebook.") for i in range(Q)]
        res = []
        for s in range(steps):
            x = xs_fn(s)
            # per-layer configs only differ in codebook size
            if len(set(sizes)) == 1:
                cfg = oracle_cfg({**kw, "codebook_size": sizes[0]})
                q, idx, loss = O.rvq_forward(states, cfg, x, shared_codebook=shared, training=meta["train"],
                                             assign_mode=mode, stats_mode=stats, **samp)
            else:
                out = torch.zeros_like(x); residual = x; ii = []; ll = []
                for st, c in zip(states, sizes):
                    cfg = oracle_cfg({**kw, "codebook_size": c})
                    qq, ind, lo = O.vq_forward(st, cfg, residual, training=meta["train"], assign_mode=mode, stats_mode=stats, **samp)
                    residual = residual - qq.detach(); out = out + qq; ii.append(ind); ll.append(lo)
                q, idx, loss = out, torch.stack(ii, -1), torch.stack(ll)
            res.append(dict(q=q, idx=idx, loss=loss))
        final = {}
        for i, st in enumerate(states):
            p = prefix + f"layers.{i}._codebook."
            final.update({p + "embed": st.embed, p + "embed_avg": st.embed_avg, p + "cluster_size": st.cluster_size})
        return res, final

    def x_of(s):        # recorded with an input that requires grad: the layers then return the ROUTED value, which rvq.py:524 subtracts
        x = fx.t(f"x{s}").clone()
        return x.requires_grad_(True) if meta["grad"] else x

    def with_grads(outs, xs):
        for s, o in enumerate(outs):
            if meta["grad"]:
                (o["loss"].sum() * 3.0 + (o["q"] * fx.t(f"gw{s}")).sum()).backward()
                o["gx"] = xs[s].grad
            o["q"], o["loss"] = o["q"].detach(), o["loss"].detach()
        return outs

    if cls == "ResidualVQ":
        xs = [x_of(s) for s in range(meta["steps"])]
        res, final = rvq(kw, sd, "", lambda s: xs[s], meta["steps"])
        return with_grads(res, xs), final
    if cls == "GroupedResidualVQ":
        G = kw["groups"]
        sub = {k: v for k, v in kw.items() if k != "groups"}
        sub["dim"] = kw["dim"] // G
        per, final = [], {}
        xs = [x_of(s) for s in range(meta["steps"])]
        for g in range(G):
            r, f = rvq(sub, sd, f"rvqs.{g}.", lambda s, g=g: xs[s].chunk(G, -1)[g], meta["steps"])
            per.append(r)
            final.update(f)
        outs = []
        for s in range(meta["steps"]):
            outs.append(dict(q=torch.cat([per[g][s]["q"] for g in range(G)], -1),
                             idx=torch.stack([per[g][s]["idx"] for g in range(G)]),
                             loss=torch.stack([per[g][s]["loss"] for g in range(G)])))
        return with_grads(outs, xs), final
    raise ValueError(cls)


def classify_mismatches(x_rows, embed2d, idx_a, idx_b, cosine=False):
    """Tie audit (SURVEY.md §7 hard part 1): for every row where two index vectors disagree, the fp32
    gap between the two candidates in the chain oracle's own score row, in ulps of the score.
    Returns list of (row, idx_a, idx_b, gap_ulps)."""
    bad = (idx_a != idx_b).nonzero().flatten().tolist()
    out = []
    for r in bad:
        sc = O.c_scores(x_rows[r:r + 1].float(), embed2d, cosine)[0]
        a, b = sc[idx_a[r]].item(), sc[idx_b[r]].item()
        ulp = abs(torch.nextafter(torch.tensor(a), torch.tensor(float("inf"))).item() - a)
        out.append((r, int(idx_a[r]), int(idx_b[r]), abs(a - b) / max(ulp, 1e-45)))
    return out
