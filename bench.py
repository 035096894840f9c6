#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ...]

A "step" is one training forward of the hot path (codebook pack + nearest-code search + gather + commit loss +
EMA statistics + EMA fold) over one synthetic batch already resident in HBM.  Default workload = BASELINE
config[1] ("vq_cfg2"): VectorQuantize(dim=256, codebook_size=1024), x = (64, 16384, 256) bf16 (2^20 vectors).

N > 1: one process per GPU over RCCL.  Launched by the driver as `python -m torch.distributed.run ... bench.py
--gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or, when WORLD_SIZE is not set, bench.py starts
those N ranks itself by re-executing under torch.distributed.run.  WORLD_SIZE must equal --gpus.  Every rank owns
its own batch (weak scaling); the EMA statistics are summed with ONE all-reduce per step, the reference's
data-parallel scheme (vqp.py:603, 607).  value = all ranks' vectors / max-over-ranks time.

Other workloads (informational, same JSON shape): rvq_cfg3, grvq_cfg5 (BASELINE configs 3 and 5 on one GPU),
vq_cfg4_shard (config 4: the work of ONE of the 8 ranks -- all 262144 rows against an 8192-code shard, D = 512,
cosine -- on one GPU) and vq_cfg4_sharded (config 4 with the codebook sharded over the N ranks, RCCL argmin merge).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

B, S, D, C = 64, 16384, 256, 1024
PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0       # same guide: dense bf16 MFMA peak (AMD's 2:1-sparse headline is not used)
PEAK_HBM_GBPS = 8000.0               # same guide: HBM3E 8 TB/s spec (6.3 TB/s achievable)
N_BATCHES = 4                        # distinct synthetic batches cycled through the timed steps (2 GiB at cfg 2)


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline + parity audit against the reference's op sequence (oracle mode "aten")
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline_and_audit(nthreads, dev):
    """Times the reference's CPU op sequence (oracle mode="aten", quantize_mode="onehot": the 3 N*C*D contractions of a training
    forward -- cdist, the one-hot gather einsum, the EMA einsum -- + the N*C temporaries; proven bit-identical to the live
    reference in the build container, tests/test_oracle.py -- /root/reference itself is not on the GPU box) on a bounded sample of the cfg-2 workload (the first 131072 vectors), then audits the GPU path's indices
    against the oracle's on ALL 2^20 rows of that batch in chunks (mismatch count with tie audit):
    for every mismatch the gap between the two candidates in the reference's own fp32 distances (ulps) and which of the two
    is closer in float64."""
    from oracle import vq_oracle as O
    from vector_quantize_pytorch_amd import _lib
    rows_b, rows_s = 8, 16384                       # 131072 vectors per oracle forward
    n_chunks = (B * S) // (rows_b * rows_s)
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(rows_b, rows_s, D, generator=g).bfloat16() for _ in range(n_chunks)]
    bound = (6.0 / (C * D)) ** 0.5
    e = (torch.rand(1, C, D, generator=g) * 2 - 1) * bound
    cfg = O.VQConfig(dim=D, codebook_size=C)

    def fresh():
        return O.VQState(embed=e.clone(), embed_avg=e.clone(), cluster_size=torch.ones(1, C))

    st = fresh()
    with torch.no_grad():
        O.vq_forward(st, cfg, xs[0], quantize_mode="onehot")                # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.vq_forward(st, cfg, xs[0], quantize_mode="onehot")
            ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    base = dict(value=rows_b * rows_s / t, unit="vectors/s", cores=nthreads, kind="port",
                sample=(f"oracle mode=aten, quantize_mode=onehot = the reference's ATen op sequence of a training forward (cdist vqp.py:58-62, "
                        f"F.one_hot(ind).type(dtype) :142 + the one-hot gather einsum :766, the EMA einsum :602-606: 3 N*C*D contractions; live "
                        f"reference not on the GPU box), x=({rows_b},{rows_s},{D}) bf16 input as in cfg 2, fp32 arithmetic (the reference casts "
                        f"at vqp.py:692), C={C}, median of 3 after 1 warm-up, {t:.3f} s/forward"))

    # ---- audit: GPU (screened + exact passes) vs the reference op sequence, same rows, same (initial) codebook ----
    ed = e[0].to(dev).contiguous()
    packed = _lib.pack_codebook(ed)
    e64 = e[0].double()
    hist = {"0": 0, "1": 0, "2": 0, ">2": 0}
    closer = {"gpu": 0, "reference": 0, "exact_tie": 0}
    n_rows = n_mism = max_ulps = 0
    with torch.no_grad():
        for x in xs:
            _, idx_aten, _ = O.vq_forward(fresh(), cfg, x)          # first-step indices on codebook e
            ia = idx_aten.reshape(-1)
            gi = _lib.assign(x.reshape(-1, D).to(dev), packed, ed, want_q=False)["idx"].cpu()
            mism = (gi != ia).nonzero().flatten()
            n_rows += ia.numel()
            n_mism += mism.numel()
            if mism.numel():
                rows = x.reshape(-1, D)[mism].float()
                d = -O.neg_cdist(rows[None], e)[0]                   # [m, C] cdist exactly as the reference computes it (fp32)
                da = d.gather(1, ia[mism][:, None])[:, 0]
                dg = d.gather(1, gi[mism][:, None])[:, 0]
                ulps = (da.view(torch.int32).long() - dg.view(torch.int32).long()).abs()
                max_ulps = max(max_ulps, int(ulps.max()))
                for u in ulps.tolist():
                    hist[str(u) if u <= 2 else ">2"] += 1
                r64 = rows.double()
                ta = ((r64 - e64[ia[mism]]) ** 2).sum(-1)
                tg = ((r64 - e64[gi[mism]]) ** 2).sum(-1)
                closer["gpu"] += int((tg < ta).sum())
                closer["reference"] += int((ta < tg).sum())
                closer["exact_tie"] += int((ta == tg).sum())
    audit = dict(rows_checked_vs_aten=n_rows, mismatches_vs_aten=n_mism, tie_audit_max_ulps=max_ulps,
                 tie_audit_ulp_histogram=hist, closer_in_float64=closer,
                 tie_audit_note=("ulps = gap between the two candidates in the reference's own fp32 cdist values; closer_in_float64 = which "
                                 "candidate is nearer when the distance is evaluated in float64 (neither arithmetic is 'right' on a 0..1-ulp gap: "
                                 "MKL's blocked sgemm and the kernels' ascending fp32 FMA chain round differently)"))
    return base, audit


def screened_vs_exact(x, vq):
    """whole-batch comparison of the screened search with the exact fp32-MFMA kernel on the current codebook"""
    from vector_quantize_pytorch_amd import _lib
    e = vq._codebook.embed[0].detach().contiguous()
    packed = _lib.pack_codebook(e)
    rows = x.reshape(-1, x.shape[-1])
    old = os.environ.get("VQHIP_SCREEN")
    try:
        os.environ["VQHIP_SCREEN"] = "1"
        r1 = _lib.assign(rows, packed, e, want_q=True)
        os.environ["VQHIP_SCREEN"] = "0"
        r0 = _lib.assign(rows, packed, e, want_q=True)
    finally:
        if old is None:
            os.environ.pop("VQHIP_SCREEN", None)
        else:
            os.environ["VQHIP_SCREEN"] = old
    bad = int((r1["idx"] != r0["idx"]).sum()) + int((r1["q"] != r0["q"]).any(-1).sum())
    return rows.shape[0], bad


# ----------------------------------------------------------------------------------------------------------------------
def _preheat(dev, seconds=0.25):
    """HBM-bound copies for `seconds` right before a timed section, so that the memory clock is at its operating state when the
    windows start.  (Two default runs of round 4 measured every forward window at 1.2 ms with the search at its usual 0.62 ms -- the
    HBM-bound statistics pass 2.6 x slower, the MFMA-bound search unchanged.  That turned out to be one box of the pool, with or
    without this loop, profiles/r4_final/bench_slow_box.json; the loop stays as cheap insurance against a memory clock that is
    still ramping when the first window starts.  Untimed, the same fixed work on every rank.)"""
    if dev.type != "cuda" or seconds <= 0:
        return
    a = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            b.copy_(a)
        torch.cuda.synchronize(dev)
    del a, b


BENCH_ENV = {"setpriority_minus10": None, "preheat_s": 0.25,
             "note": "round 4+: the issuing process asks for nice -10 (needs CAP_SYS_NICE: `setpriority_minus10` says whether it was granted) and "
                     "0.25 s of untimed HBM copies precede every timed section; earlier rounds' numbers were taken without either"}


def _dist_facts(world, dev):
    """what a SCALE record needs to prove N ranks took part: ranks_seen = an all-reduce of ones over the process group, the RCCL
    version torch was built against"""
    out = {"ranks_seen": 1, "rccl_version": None}
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    if world > 1:
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        out["ranks_seen"] = int(one.item())
    return out


def _windows(run_step, steps, windows, sync):
    """`windows` back-to-back timed windows of exactly `steps` steps, each bracketed by barrier + synchronize on both sides;
    returns the per-window wall times (the reported step time is the MEDIAN window: one 20-step window is ~20 ms at cfg 2)"""
    dts = []
    k = 0
    for _ in range(windows):
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step(k)
            k += 1
        sync()
        dts.append(time.perf_counter() - t0)
    return dts


def _mean(v):
    return None if not v else round(sum(v) / len(v), 5)


def _median(v):
    return sorted(v)[len(v) // 2]


def _time_module(mod, batches, steps, warmup, sync, windows, after_first=None):
    out = [None]
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mod(batches[0])                                 # first forward (k-means init for cfg 5)
        torch.cuda.synchronize(); first = time.perf_counter() - t0
        if after_first is not None:
            after_first()
        for i in range(warmup):
            mod(batches[i % len(batches)])

        def step(k):
            out[0] = mod(batches[k % len(batches)])
        _preheat(batches[0].device)
        dts = _windows(step, steps, windows, sync)
    return dts, first, out[0]


def _time_grad_step(mod, batches, steps, warmup, sync, windows):
    """BASELINE.md §4's second line: the same module and batches, `x.requires_grad_()` + backward with a resident upstream gradient for `quantized` plus the summed commitment loss(es):
    training forward (search, EMA update) plus the gradient to the input through the straight-through / rotation-trick route and
    the commitment loss."""
    xs = [b.clone().requires_grad_(True) for b in batches]
    gq = torch.randn_like(batches[0])               # the upstream gradient of `quantized` (a decoder's), resident like the batches

    def step(k):
        x = xs[k % len(xs)]
        x.grad = None
        res = mod(x)
        torch.autograd.backward((res[0], res[2].sum()), (gq, None))     # dL/dx = J^T gq + d(sum of the commit losses)/dx
    for i in range(max(warmup, 1)):
        step(i)
    _preheat(batches[0].device)
    return _windows(step, steps, windows, sync)


def cpu_baseline_other(workload, nthreads):
    """the reference's op sequence for configs 3 / 4 / 5 timed on a ROW CHUNK of the workload on the host cores and
    scaled (throughput is flat in the row count at these sizes): oracle mode="aten" with the one-hot gather of a training forward
    (quantize_mode="onehot") for the residual modules, eval mode for cfg 4 as BASELINE.md prescribes.  Steady state (codebooks marked
    initialised: the k-means first forward of cfg 5 is not part of a step).  -> dict for the JSON line."""
    from oracle import vq_oracle as O
    g = torch.Generator().manual_seed(0)
    if workload == "rvq_cfg3":
        rows, Dm, Cm, Q, G = (2, 8192), 256, 1024, 8, 1
    elif workload == "grvq_cfg5":
        rows, Dm, Cm, Q, G = (1, 4096), 512, 4096, 8, 4
    else:
        rows, Dm, Cm, Q, G = (1, 4096), 512, 65536, 1, 1
    n = rows[0] * rows[1]
    x = torch.randn(*rows, Dm, generator=g)
    d = Dm // G

    def state(cosine=False):
        e = (torch.rand(1, Cm, d, generator=g) * 2 - 1) * (6.0 / (Cm * d)) ** 0.5
        if cosine:
            e = torch.nn.functional.normalize(e, dim=-1)
        return O.VQState(embed=e.clone(), embed_avg=e.clone(), cluster_size=torch.ones(1, Cm), initted=True)

    if workload in ("rvq_cfg3", "grvq_cfg5"):
        shared = workload == "rvq_cfg3"
        cfg = O.VQConfig(dim=d, codebook_size=Cm, manual_ema_update=shared)
        groups = []
        for _ in range(G):
            sts = [state()] * Q if shared else [state() for _ in range(Q)]
            groups.append(sts)

        def fwd():
            for gi, sts in enumerate(groups):
                O.rvq_forward(sts, cfg, x[..., gi * d:(gi + 1) * d], shared_codebook=shared, quantize_mode="onehot")
        what = "train forward"
    else:
        cfg = O.VQConfig(dim=d, codebook_size=Cm, use_cosine_sim=True)
        st = state(cosine=True)

        def fwd():
            O.vq_forward(st, cfg, x, training=False)
        what = ("eval forward against ALL 65536 codes (BASELINE.md 4: cfg 4 in eval mode; the GPU line beside it is ONE of 8 ranks' share: "
                "every row against 8192 codes)")
    with torch.no_grad():
        fwd()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fwd()
            ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    return dict(value=n / t, unit="vectors/s", cores=nthreads, kind="port",
                sample=(f"oracle mode=aten (the reference's ATen op sequence; live reference not on the GPU box), {what} on a row chunk "
                        f"x=({rows[0]},{rows[1]},{Dm}) fp32 of the workload, scaled per vector; median of 3 after 1 warm-up, {t:.3f} s/forward"))


def other_workload(args, world, rank, dev, workload=None, steps=None, warmup=None, windows=None):
    """BASELINE configs 3, 4 and 5 (informational; the contract line is vq_cfg2).  Returns the JSON line as a dict (rank 0) or None."""
    class _A:        # the same measurement under other step counts (the compact `other_workloads` object of the default line)
        pass
    a2 = _A()
    a2.__dict__.update(vars(args))
    a2.workload = workload or args.workload
    a2.steps, a2.warmup, a2.windows = steps or args.steps, (args.warmup if warmup is None else warmup), windows or args.windows
    args = a2
    from vector_quantize_pytorch_amd import GroupedResidualVQ, ResidualVQ, _lib
    from vector_quantize_pytorch_amd.parallel import ShardedVectorQuantize
    torch.manual_seed(0)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    par = "single GPU"
    if args.workload == "rvq_cfg3":
        mod = ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True).to(dev).train()
        shape, stages, flops = (32, 8192, 256), 8, 2.0 * 32 * 8192 * 8 * 1024 * 256
        name = "ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True) train forward, x=(32,8192,256) fp32"
    elif args.workload == "grvq_cfg5":
        mod = GroupedResidualVQ(dim=512, groups=4, num_quantizers=8, codebook_size=4096, kmeans_init=True).to(dev).train()
        shape, stages, flops = (32, 8192, 512), 32, 2.0 * 32 * 8192 * 4 * 8 * 4096 * 128
        name = "GroupedResidualVQ(dim=512, groups=4, num_quantizers=8, codebook_size=4096, kmeans_init=True) steady-state train forward, x=(32,8192,512) fp32"
    elif args.workload == "vq_cfg4_shard":
        # the per-rank work of config 4 on 8 GPUs: every row of the batch against this rank's 8192 of the 65536 codes
        mod = ShardedVectorQuantize(512, 65536, use_cosine_sim=True, emulate=(0, 8)).to(dev).train()
        shape, stages, flops = (16, 16384, 512), 1, 2.0 * 16 * 16384 * 8192 * 512
        name = ("VectorQuantize(dim=512, codebook_size=65536, use_cosine_sim=True) sharded over 8 ranks: ONE rank's work "
                "(all 262144 gathered rows x its 8192 codes: search, exact winner scores, decode + EMA of the codes it owns; l2norm, "
                "outputs and loss for its own 32768 rows; collectives excluded), x=(16,16384,512) fp32")
    else:   # vq_cfg4_sharded: the whole of config 4 over the `world` ranks
        mod = ShardedVectorQuantize(512, 65536, use_cosine_sim=True).to(dev).train()
        assert 16 % world == 0
        shape, stages, flops = (16 // world, 16384, 512), 1, 2.0 * 16 * 16384 * 65536 * 512 / world
        name = (f"VectorQuantize(dim=512, codebook_size=65536, use_cosine_sim=True), codebook sharded over {world} rank(s), "
                f"x=(16,16384,512) fp32 in total ({16 // world} x 16384 rows per rank, all-gathered), RCCL MAX all-reduce of packed (score, index) keys")
        par = (f"codebook sharded x{world} (rows all-gathered, one int64 MAX all-reduce, then the cheaper of: all-gather of the codebook "
               f"shards + local decode / reduce-scatter of the decoded rows)")
    nb = 2
    batches = [torch.randn(*shape, generator=gen, device=dev) for _ in range(nb)]
    if args.workload == "vq_cfg4_shard":     # what the all-gather hands a rank: rows its peers have already normalised
        batches = [torch.nn.functional.normalize(b, dim=-1) for b in batches]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # rows the screen could not certify, per search (stage): device-side counters collected without synchronising
    counts = []
    orig_assign = _lib.assign

    def counting_assign(*a, **k):
        r = orig_assign(*a, **k)
        if r.get("n_exact") is not None:
            counts.append((r["n_exact"], r["n_pair"], r["idx"].numel()))
        return r

    orig_chained = _lib.rvq_forward_chained

    def counting_chained(*a, **k):          # the chained residual loop reports its per-stage counters itself
        r = orig_chained(*a, **k)
        n = r["idx"].numel() // len(r["counts"])
        counts.extend((c[0], c[1], n) for c in r["counts"])
        return r

    _lib.assign = counting_assign
    _lib.rvq_forward_chained = counting_chained
    import vector_quantize_pytorch_amd.codebook as cbmod
    cbmod.L.assign = counting_assign
    first_counts = []

    def after_first():          # everything counted so far belongs to the FIRST forward (cfg 5: the k-means iterations, vqp.py:238-278)
        first_counts.extend(counts)
        counts.clear()

    dts, first, _ = _time_module(mod, batches, args.steps, args.warmup, sync, args.windows, after_first=after_first)
    _lib.assign = orig_assign
    _lib.rvq_forward_chained = orig_chained
    cbmod.L.assign = orig_assign
    # steady state: the counters of the LAST timed forward.  The residual loops run as one native call (vqhip_rvq_chain_forward), which
    # leaves its per-stage device counters on the module (`last_counts`: per stage (open rows, pair rows), one counter per row chunk
    # [and group]); searches issued through L.assign (the sharded module) are in `counts`.
    n_rows = shape[0] * shape[1]
    per_stage = None
    holders = [mod] if getattr(mod, "last_counts", None) is not None else [r for r in getattr(mod, "rvqs", []) if getattr(r, "last_counts", None) is not None]
    if holders:
        op, pr = [], []
        for h in holders:                                # (group-major for a grouped module on side streams)
            for c in h.last_counts:
                o, p_ = c[0].double().reshape(c[0].shape[0], -1).sum(0), c[1].double().reshape(c[1].shape[0], -1).sum(0)    # sum over the row chunks
                op += [round(float(v) / n_rows, 5) for v in o.tolist()]
                pr += [round(float(v) / n_rows, 5) for v in p_.tolist()]
        per_stage = {"open_frac": op, "pair_frac": pr, "order": "stage-major, then group" if holders == [mod] and hasattr(mod, "rvqs") else "group-major, then stage",
                     "source": "device counters of the last timed forward (vqhip_rvq_chain_forward workspace headers)"}
    elif counts:
        n_last = stages if len(counts) >= stages else len(counts)
        last = counts[-n_last:]                          # the searches of the last forward, in call order
        per_stage = {"open_frac": [round(float(c[0].sum().item()) / c[2], 5) for c in last],      # (.sum(): one counter per row chunk)
                     "pair_frac": [round(float(c[1].sum().item()) / c[2], 5) for c in last],
                     "source": "device counters of the last timed forward's L.assign calls"}
    kmeans_open = None
    if first_counts and args.workload == "grvq_cfg5":
        kmeans_open = {"open_frac_mean": _mean([float(c[0].sum().item()) / c[2] for c in first_counts]),
                       "open_frac_iteration0_mean": _mean([float(c[0].sum().item()) / c[2] for c in first_counts[0::10]]),
                       "pair_frac_mean": _mean([float(c[1].sum().item()) / c[2] for c in first_counts]),
                       "searches": len(first_counts),
                       "note": "the searches of the FIRST forward only: 10 k-means iterations per codebook (vqp.py:238-278); not part of a steady-state step"}
    gdts = None
    if args.workload in ("rvq_cfg3", "grvq_cfg5") and not args.no_grad_step:
        gdts = _time_grad_step(mod, batches, args.steps, args.warmup, sync, args.windows)
    tmax = torch.tensor(dts, dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dts = tmax.tolist()
    dt = _median(dts)
    if rank != 0:
        return
    strong = args.workload == "vq_cfg4_sharded"
    n = shape[0] * shape[1] * (world if strong else 1)
    screened = _lib.screening_enabled()
    peak = PEAK_BF16_MFMA_TFLOPS if screened else PEAK_FP32_MFMA_TFLOPS
    ach = flops * args.steps / dt / 1e12
    traffic = traffic_src = None
    try:        # HBM bytes of one step from the committed rocprofv3 PMC passes of this workload (tools/collect_profile.py), not re-measured here
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("step_traffic", {}).get(args.workload)
        if tj and screened:
            traffic, traffic_src = tj["bytes_per_step"], tj.get("source")
    except Exception:
        pass
    cpu_base = eager = None
    if world == 1 and not args.no_cpu_baseline and args.workload in ("rvq_cfg3", "grvq_cfg5", "vq_cfg4_shard"):
        cpu_base = cpu_baseline_other(args.workload, torch.get_num_threads())
        if args.workload == "rvq_cfg3":
            try:
                eager = eager_rocm_rvq3(dev, batches[0])
                eager["speedup_of_this_library"] = round(n * args.steps / dt / eager["value"], 2)
            except Exception as ex:
                eager = {"error": f"{type(ex).__name__}: {ex}"}
    return ({"metric": "vectors quantized/sec", "value": n * args.steps / dt, "unit": "vectors/s", "n_gpus": world,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
                      "scaling": "strong" if strong else "weak", "vs_baseline": None,
                      "dtype": "f16+f32" if screened else "f32", "data": "synthetic",
                      "windows_ms_per_step": [round(d / args.steps * 1e3, 4) for d in dts],
                      "grad_step": None if gdts is None else {
                          "workload": "same module and batches, x.requires_grad_() + backward (upstream gradient of `quantized` resident in HBM, + sum of the commit losses) (BASELINE.md §4, second line)",
                          "ms_per_step": _median(gdts) / args.steps * 1e3, "value": n * args.steps / _median(gdts), "unit": "vectors/s",
                          "windows_ms_per_step": [round(d / args.steps * 1e3, 4) for d in gdts]},
                      "cpu_baseline": cpu_base, "eager_rocm": eager,
                      "config": {"workload": name, "parallelism": par, "vector_stages_per_s": n * stages * args.steps / dt,
                                 "world_size": world, "backend": (dist.get_backend() if world > 1 else None),
                                 **_dist_facts(world, dev), "bench_env": BENCH_ENV,
                                 "collective_bytes_per_rank_and_step": getattr(mod, "last_comm", None) or None,
                                 "first_forward_ms": first * 1e3, "uncertified_rows_per_search": per_stage, "kmeans_first_forward": kmeans_open},
                      "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                   "traffic": traffic, "traffic_source": traffic_src,
                                   "note": "whole step (all kernels) PER GPU, not one kernel; algorithmic flops (2*C*D per vector and stage); "
                                           "peak = dense f16 MFMA when the search runs screened (VQHIP_SCREEN != 0), fp32 MFMA otherwise"}})


# ----------------------------------------------------------------------------------------------------------------------
def exact_kernel_leg(x, vq, launches=3):
    """The exact fp32-MFMA search (vq_assign_kernel, VQHIP_SCREEN=0) on one cfg-2 batch and the current codebook, HIP-event timed on
    the launch stream: the arithmetic the screened search defers to, priced against the fp32-MFMA peak."""
    from vector_quantize_pytorch_amd import _lib
    e = vq._codebook.embed[0].detach().contiguous()
    packed = _lib.pack_codebook(e)
    rows = x.reshape(-1, x.shape[-1])
    old = os.environ.get("VQHIP_SCREEN")
    ts = []
    try:
        os.environ["VQHIP_SCREEN"] = "0"
        _lib.assign(rows, packed, e, want_q=True)
        for _ in range(launches):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.assign(rows, packed, e, want_q=True)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    finally:
        if old is None:
            os.environ.pop("VQHIP_SCREEN", None)
        else:
            os.environ["VQHIP_SCREEN"] = old
    ms = _median(ts)
    tf = 2.0 * rows.shape[0] * e.shape[0] * e.shape[1] / (ms * 1e-3) / 1e12
    return {"kernel": "vq_assign_kernel<256,bf16,euclid> (VQHIP_SCREEN=0: index + q outputs)", "ms": ms, "launches": launches,
            "achieved_tflops": tf, "frac_of_fp32_mfma_peak": tf / PEAK_FP32_MFMA_TFLOPS}


def eager_rocm_baseline(dev, batch):
    """BASELINE.md 4, secondary datapoint: what a user of the reference gets TODAY on this GPU -- the reference's own op sequence
    (oracle mode "aten", quantize_mode "onehot": cdist vqp.py:58-62, argmax :140, F.one_hot + gather einsum :142 / :766, EMA einsums
    :602-606, lerps :76-97, Laplace :152-154) run eagerly by PyTorch-ROCm on `cuda` tensors (hipBLASLt / rocBLAS contractions + the
    N x C temporaries), at cfg 2's full size, HIP-event timed.  Same arithmetic the CPU baseline times; a baseline, not the product."""
    from oracle import vq_oracle as O
    g = torch.Generator().manual_seed(0)
    bound = (6.0 / (C * D)) ** 0.5
    e = ((torch.rand(1, C, D, generator=g) * 2 - 1) * bound).to(dev)
    st = O.VQState(embed=e.clone(), embed_avg=e.clone(), cluster_size=torch.ones(1, C, device=dev))
    cfg = O.VQConfig(dim=D, codebook_size=C)
    ts = []
    torch.cuda.reset_peak_memory_stats(dev)
    with torch.no_grad():
        O.vq_forward(st, cfg, batch, quantize_mode="onehot")
        torch.cuda.synchronize()
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            O.vq_forward(st, cfg, batch, quantize_mode="onehot")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    ms = _median(ts)
    n = batch.shape[0] * batch.shape[1]
    peak = torch.cuda.max_memory_allocated(dev)
    torch.cuda.empty_cache()
    return {"value": n / (ms * 1e-3), "unit": "vectors/s", "ms_per_step": ms, "kind": "port",
            "peak_allocated_gb": round(peak / 2**30, 2),
            "sample": (f"oracle mode=aten, quantize_mode=onehot (the reference's ATen op sequence of a training forward, proven bit-identical to the "
                       f"live reference on CPU; live reference not on the GPU box) on cuda tensors through PyTorch-ROCm, x=({B},{S},{D}) bf16 = cfg 2 at "
                       f"full size, C={C}, median of 3 after 1 warm-up, HIP events")}


def eager_rocm_rvq3(dev, batch):
    """the reference's op sequence of a cfg-3 training forward (oracle rvq_forward, mode "aten", one-hot gather) eagerly through
    PyTorch-ROCm on cuda tensors, full size (32 x 8192 rows, 8 stages, one shared codebook), HIP-event timed"""
    from oracle import vq_oracle as O
    g = torch.Generator().manual_seed(0)
    Cm, Dm, Q = 1024, 256, 8
    e = ((torch.rand(1, Cm, Dm, generator=g) * 2 - 1) * (6.0 / (Cm * Dm)) ** 0.5).to(dev)
    st = O.VQState(embed=e.clone(), embed_avg=e.clone(), cluster_size=torch.ones(1, Cm, device=dev), initted=True)
    cfg = O.VQConfig(dim=Dm, codebook_size=Cm, manual_ema_update=True)
    ts = []
    with torch.no_grad():
        O.rvq_forward([st] * Q, cfg, batch, shared_codebook=True, quantize_mode="onehot")
        torch.cuda.synchronize()
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            O.rvq_forward([st] * Q, cfg, batch, shared_codebook=True, quantize_mode="onehot")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    ms = _median(ts)
    torch.cuda.empty_cache()
    return {"value": batch.shape[0] * batch.shape[1] / (ms * 1e-3), "unit": "vectors/s", "ms_per_step": ms, "kind": "port",
            "sample": "oracle rvq_forward mode=aten, quantize_mode=onehot on cuda tensors through PyTorch-ROCm, cfg 3 at full size, median of 3 after 1 warm-up, HIP events"}


def screen_stress(dev, base_ms):
    """the screened train step away from its friendliest input (randn rows, default init).  Same module, shapes and
    dtype as cfg 2; every leg reports ms per step and the uncertified fractions of its last steps.
      all_open        every code has two identical twins: best, second and third tie for every row -> every row takes the exact fp32-MFMA
                      sweep (vq_refine_kernel): the floor of the search
      pair_floor      every code has ONE twin: every row is decided between two codes by two exact distances (vq_pair_kernel)
      gmm_normspread  rows from a 64-component Gaussian mixture whose components' scales span 30 x, codebook trained on it for 50 EMA steps
      one_code_x100   cfg 2's default codebook with ONE code scaled x 100 (its norm feeds the certificate's codebook-side terms)"""
    from vector_quantize_pytorch_amd import VectorQuantize
    import vector_quantize_pytorch_amd.codebook as cbmod
    out = {}
    gen = torch.Generator(device=dev).manual_seed(7)
    rand_batches = [torch.randn(B, S, D, generator=gen, device=dev).bfloat16() for _ in range(2)]

    def run(vq, batches, steps, prep=None, warm=1):
        fr = []
        orig = cbmod.L.vq_train_step

        def counting(*a, **k):
            r = orig(*a, **k)
            fr.append((r["n_exact"][0].clone(), r["n_pair"][0].clone()))
            return r
        with torch.no_grad():
            for i in range(warm):
                if prep:
                    prep(vq)
                vq(batches[i % len(batches)])
            wins = []                               # three windows of `steps`, the median (a one-off host stall inside a short
            for w in range(3):                      # window -- an allocator miss, a code-object load -- was 2 - 10 ms per step)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(steps):
                    if prep:
                        prep(vq)
                    vq(batches[i % len(batches)])
                torch.cuda.synchronize(); wins.append((time.perf_counter() - t0) / steps * 1e3)
            ms = sorted(wins)[1]
            cbmod.L.vq_train_step = counting
            try:
                for i in range(2):
                    if prep:
                        prep(vq)
                    vq(batches[i % len(batches)])
            finally:
                cbmod.L.vq_train_step = orig
            torch.cuda.synchronize()
        n = B * S
        return {"ms_per_step": round(ms, 4), "vs_cfg2": None if not base_ms else round(ms / base_ms, 3),
                "open_frac": _mean([float(a.item()) / n for a, _ in fr]), "pair_frac": _mean([float(b.item()) / n for _, b in fr]), "steps": steps}

    def twins(k):
        def prep(vq):
            cb = vq._codebook
            m = C // k
            for t in range(1, k):
                cb.embed[0, t * m:(t + 1) * m] = cb.embed[0, :m]
        return prep

    torch.manual_seed(0)
    out["all_open"] = run(VectorQuantize(dim=D, codebook_size=C).to(dev).train(), rand_batches, 3, prep=twins(3))
    out["all_open"]["workload"] = "cfg-2 step, every code with two identical twins (3 x 341 codes): 100 % of the rows take the exact sweep"
    out["pair_floor"] = run(VectorQuantize(dim=D, codebook_size=C).to(dev).train(), rand_batches, 3, prep=twins(2))
    out["pair_floor"]["workload"] = "cfg-2 step, every code with one identical twin: 100 % of the rows are decided between two codes (vq_pair_kernel)"
    # Gaussian mixture, 64 components, scales 1 .. 30 (log-spaced): x = (mu_k + 0.5 z) s_k
    K = 64
    mu = torch.randn(K, D, generator=gen, device=dev)
    sk = torch.logspace(0, torch.log10(torch.tensor(30.0)).item(), K, device=dev)
    gmm = []
    for _ in range(2):
        comp = torch.randint(0, K, (B, S), generator=gen, device=dev)
        gmm.append(((mu[comp] + 0.5 * torch.randn(B, S, D, generator=gen, device=dev)) * sk[comp][..., None]).bfloat16())
    vq = VectorQuantize(dim=D, codebook_size=C).to(dev).train()
    out["gmm_normspread"] = run(vq, gmm, 10, warm=50)
    out["gmm_normspread"]["workload"] = ("cfg-2 step on rows from a 64-component Gaussian mixture, component scales 1 .. 30 (row norms span 30 x), codebook "
                                         "trained on it for 50 EMA steps first")
    e2 = (vq._codebook.embed[0] ** 2).sum(-1)
    out["gmm_normspread"]["codebook_norm2_min_max"] = [float(e2.min()), float(e2.max())]
    del gmm
    vq = VectorQuantize(dim=D, codebook_size=C).to(dev).train()
    with torch.no_grad():
        vq._codebook.embed[0, 7] *= 100.0
        vq._codebook.embed_avg[0, 7] *= 100.0

    def keep_big(v):        # (the EMA keeps pulling the code towards the rows it wins: hold it at x 100 of a default code's size)
        cb = v._codebook
        n7 = cb.embed[0, 7].norm()
        cb.embed[0, 7] *= (100.0 * cb.embed[0, 8].norm() / n7.clamp_min(1e-20)).clamp(max=1e6)
    # control: the same module and per-step rescaling launches with the code held at 1 x (what the leg costs without the large code).
    # The two are timed alternately, three turns each, the median turn: a 10-step leg timed once carried its own warm-up.
    vq_ctl = VectorQuantize(dim=D, codebook_size=C).to(dev).train()

    def keep_one(v):
        cb = v._codebook
        n7 = cb.embed[0, 7].norm()
        cb.embed[0, 7] *= (1.0 * cb.embed[0, 8].norm() / n7.clamp_min(1e-20)).clamp(max=1e6)
    big, ctl = [], []
    for turn in range(3):
        big.append(run(vq, rand_batches, 10, prep=keep_big, warm=2))
        ctl.append(run(vq_ctl, rand_batches, 10, prep=keep_one, warm=2))
    big.sort(key=lambda r: r["ms_per_step"]); ctl.sort(key=lambda r: r["ms_per_step"])
    out["one_code_x100"] = big[1]
    out["one_code_x100"]["workload"] = "cfg-2 step, default codebook with code 7 held at 100 x the norm of its neighbour (three turns alternating with the control, each the median of three 10-step windows; the median turn)"
    out["one_code_x100"]["control_ms_per_step"] = ctl[1]["ms_per_step"]
    out["one_code_x100"]["vs_control"] = round(big[1]["ms_per_step"] / ctl[1]["ms_per_step"], 3)
    return out


# ----------------------------------------------------------------------------------------------------------------------
def vq_cfg2(args, world, rank, dev):
    from vector_quantize_pytorch_amd import VectorQuantize, _lib
    import vector_quantize_pytorch_amd.codebook as cbmod

    torch.manual_seed(0)
    vq = VectorQuantize(dim=D, codebook_size=C, sync_codebook=(world > 1)).to(dev).train()
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    batches = [torch.randn(B, S, D, generator=gen, device=dev).bfloat16() for _ in range(N_BATCHES)]   # resident in HBM

    # per-launch timing of the dominant work (the nearest-code search) with events on the launch stream (= torch's current
    # stream: the library launches on the stream it is handed)
    ev, exact_rows, pair_rows = [], [], []
    orig_assign = _lib.assign

    def timed_assign(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig_assign(*a, **k)
        e1.record()
        ev.append((e0, e1))
        if r.get("n_exact") is not None:
            exact_rows.append(r["n_exact"][0])
            pair_rows.append(r["n_pair"][0])
        return r

    cbmod.L.assign = timed_assign
    # the fused train step (vqhip_vq_train_step) contains the search: the library records the two events around it itself
    orig_step = _lib.vq_train_step

    pool = []                                        # event pairs made (handles created) ahead of the timed windows

    def make_pair():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()                     # (creates the handles; the library records them again on the same stream)
        return e0, e1

    def event_pair():
        pr = pool.pop() if pool else make_pair()
        ev.append(pr)
        return pr

    counting = [True]

    def counting_step(*a, **k):
        r = orig_step(*a, **k)
        if counting[0]:          # (views of the step's reused workspace: snapshot them -- outside the timed windows only)
            exact_rows.append(r["n_exact"][0].clone())
            pair_rows.append(r["n_pair"][0].clone())
        return r

    _lib.step_event_hook = event_pair
    cbmod.L.vq_train_step = counting_step

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        vq(batches[0])                               # step 1 on the reference's default init
        sync()
        first_exact = float(exact_rows[0].item()) / (B * S) if exact_rows else None
        first_pair = float(pair_rows[0].item()) / (B * S) if pair_rows else None
        for i in range(max(args.warmup - 1, 0)):
            vq(batches[(i + 1) % N_BATCHES])
        sync()
        ev.clear(); exact_rows.clear(); pair_rows.clear()
        pool.extend(make_pair() for _ in range(args.steps * args.windows))      # inside the windows only the library's two records run
        sync()
        last = [None]

        def step(k):
            last[0] = vq(batches[k % N_BATCHES])
        _preheat(dev)
        counting[0] = False
        dts = _windows(step, args.steps, args.windows, sync)
        q, idx, loss = last[0]
        counting[0] = True          # the uncertified-row counters of one more pass over the batches (untimed, no events)
        _lib.step_event_hook = None
        for k in range(N_BATCHES):
            vq(batches[k])
        sync()

    cbmod.L.assign = orig_assign
    cbmod.L.vq_train_step = orig_step
    _lib.step_event_hook = None
    gdts = None
    if not args.no_grad_step:
        gdts = _time_grad_step(vq, batches, args.steps, args.warmup, sync, args.windows)
    tmax = torch.tensor(dts + (gdts or []), dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tl = tmax.tolist()
    dts, gdts = tl[: len(dts)], (tl[len(dts):] if gdts else None)
    dt = _median(dts)
    dist_facts = _dist_facts(world, dev)            # (collective: every rank)
    if rank != 0:
        return

    n_vec = B * S
    k_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    win_search = None
    if len(ev) == args.steps * args.windows:        # one event pair per step, in step order
        per = [a.elapsed_time(b) for a, b in ev]
        win_search = [round(sum(per[w * args.steps:(w + 1) * args.steps]) / args.steps, 4) for w in range(args.windows)]
    flops = 2.0 * n_vec * C * D                                  # 2*C*D per vector
    alg_bytes = n_vec * 1032 + C * D * 4                         # D*2 in + D*2 out + 8 per vector (+ codebook once)
    achieved = flops / (k_ms * 1e-3) / 1e12
    screened = _lib.screening_enabled()
    traffic, traffic_src, step_traffic = None, None, None
    tp = os.path.join(ROOT, "profiles", "traffic.json")          # HBM bytes per launch from rocprofv3 PMC passes (not measured in this run)
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            traffic = tj.get("assign_screened_cfg2_bytes_per_launch" if screened else "vq_assign_kernel_cfg2_bytes_per_launch")
            traffic_src = ("profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; measured at commit "
                           f"{(tj.get('assign_screened_cfg2') or {}).get('measured_at_commit', 'n/a')}, not re-measured in this run)")
            step_traffic = (tj.get("step_traffic", {}).get("vq_cfg2") or {}).get("bytes_per_step") if screened else None
        except Exception:
            traffic = None
    peak = PEAK_BF16_MFMA_TFLOPS if screened else PEAK_FP32_MFMA_TFLOPS
    step_s = dt / args.steps
    out = {
        "metric": "vectors quantized/sec (VectorQuantize train forward, dim=256 cb=1024)",
        "value": world * n_vec * args.steps / dt,
        "unit": "vectors/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16+f32" if screened else "f32",
        "data": "synthetic",
        "windows_ms_per_step": [round(d / args.steps * 1e3, 4) for d in dts],   # ms_per_step / value = the median window
        # per window: the search of a step (events recorded by the library around it) and everything else of the step (statistics, fold,
        # pack, launch gaps) -- so that a slow window can be attributed
        "windows_search_ms": win_search,
        "windows_rest_ms": None if win_search is None else [round(d / args.steps * 1e3 - sm, 4) for d, sm in zip(dts, win_search)],
        "grad_step": None if gdts is None else {
            "workload": ("same module and batches, x.requires_grad_() + backward (upstream gradient of `quantized` resident in HBM, + the commit loss): rotation-trick route (the "
                         "module default, vqp.py:856) + commit-loss gradient (BASELINE.md §4, second line)"),
            "ms_per_step": _median(gdts) / args.steps * 1e3, "value": world * n_vec * args.steps / _median(gdts), "unit": "vectors/s",
            "windows_ms_per_step": [round(d / args.steps * 1e3, 4) for d in gdts]},
        "hbm_gbps": alg_bytes / step_s / 1e9,                     # algorithmic bytes of a step / step time, per GPU
        "hbm_frac": alg_bytes / step_s / 1e9 / PEAK_HBM_GBPS,
        "config": {"workload": f"VectorQuantize(dim={D}, codebook_size={C}) train forward + EMA update, x=({B},{S},{D}) bf16 per GPU, "
                               f"{N_BATCHES} distinct batches cycled, codebook evolving by EMA",
                   "vectors_per_gpu": n_vec,
                   "parallelism": f"dp{world} (rows sharded, one all-reduce of EMA statistics per step)" if world > 1 else "single GPU",
                   "world_size": world, "backend": (dist.get_backend() if world > 1 else None),
                   **dist_facts, "bench_env": BENCH_ENV, "loss": float(loss.item())},
        "roofline": {"bound": "mfma",
                     "kernel": (("vq_screenc_kernel<256> + vq_compact_lists_kernel" if os.environ.get("VQHIP_SCREEN_PERSIST", "1") != "0" else "vq_screen16_kernel<256>")
                                + " + vq_refine_kernel<256> + vq_pair_kernel<256> + vq_finish_listed_kernel" if screened
                                else "vq_assign_kernel<256,bf16,euclid>"),
                     "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": traffic, "traffic_source": traffic_src, "traffic_whole_step": step_traffic,
                     "kernel_ms": k_ms, "algorithmic_flops_per_launch": flops,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "note": ("achieved counts algorithmic flops (2*C*D per vector) over the whole search (screen + exact pass on the "
                              "uncertified rows + finish), timed with events on the launch stream") if screened else
                             "exact fp32-MFMA search (VQHIP_SCREEN=0)"},
    }
    if screened:
        y2 = (vq._codebook.embed[0].double() ** 2).sum(-1)
        ratio = float(y2.max() / y2.min().clamp_min(1e-300))
        out["roofline"]["codebook_norm2_max_over_min"] = ratio
        out["roofline"]["certificate_mode"] = ("plain scores, per-code allowances for winner and runner-up (norm spread <= 4 x)" if ratio <= 16.0
                                               else "upper-bound scores (per-code allowance inside every start value)")
    if screened and exact_rows:
        out["roofline"]["rows_exact_pass_frac"] = float(torch.stack(exact_rows).double().mean().item()) / n_vec
        out["roofline"]["rows_exact_pass_frac_first_step"] = first_exact
        out["roofline"]["rows_pair_pass_frac"] = float(torch.stack(pair_rows).double().mean().item()) / n_vec
        out["roofline"]["rows_pair_pass_frac_first_step"] = first_pair
    if world == 1:
        parity = {}
        if screened:
            n_chk, bad = screened_vs_exact(batches[0], vq)
            parity.update(rows_checked=n_chk, mismatches_vs_exact=bad)
            out["roofline"]["exact_kernel"] = exact_kernel_leg(batches[0], vq)
        if not args.no_adversarial and screened:
            # throughput floor: a codebook in which every code has an identical twin -- no row can be certified, every row
            # takes the screen AND the exact pass
            dup = VectorQuantize(dim=D, codebook_size=C).to(dev).train()
            with torch.no_grad():
                cb = dup._codebook
                cb.embed[0, C // 2:] = cb.embed[0, : C // 2]
                cb.embed_avg.copy_(cb.embed)
                k = 3
                dup(batches[0])
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(k):
                    cb.embed[0, C // 2:] = cb.embed[0, : C // 2]          # keep the twins identical across the EMA update
                    dup(batches[i % N_BATCHES])
                torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / k
            out["adversarial"] = {"workload": ("PAIR-PATH floor: same step with a duplicated codebook (every code has one twin): 100 % of the rows are decided "
                                               "between two codes by vq_pair_kernel.  The all-open floor (every row through the exact sweep) is "
                                               "screen_stress.all_open"),
                                  "ms_per_step": ta * 1e3, "value": n_vec / ta, "unit": "vectors/s"}
            try:
                out["screen_stress"] = screen_stress(dev, step_s * 1e3)
            except Exception as ex:          # informational legs must not take the contract line down
                out["screen_stress"] = {"error": f"{type(ex).__name__}: {ex}"}
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            base, audit = cpu_baseline_and_audit(torch.get_num_threads(), dev)
            out["cpu_baseline"] = base
            parity.update(audit)
            try:
                out["eager_rocm"] = eager_rocm_baseline(dev, batches[0])
                out["eager_rocm"]["speedup_of_this_library"] = round(out["value"] / world / out["eager_rocm"]["value"], 2)
            except Exception as ex:
                out["eager_rocm"] = {"error": f"{type(ex).__name__}: {ex}"}
        out["parity"] = parity
        if not args.no_other_workloads:
            # BASELINE configs 3 / 4 (one rank's shard work) / 5 under the driver's clock as well: the same measurement as
            # `--workload X`, fewer steps, compacted (a few seconds in total)
            del batches, q, idx, loss, last
            vq = None
            torch.cuda.empty_cache()
            ow = {}
            for wl, st in (("rvq_cfg3", 10), ("grvq_cfg5", 5), ("vq_cfg4_shard", 10)):
                try:
                    r = other_workload(args, 1, 0, dev, workload=wl, steps=st, warmup=2, windows=3)
                    ow[wl] = {"ms_per_step": round(r["ms_per_step"], 4), "vectors_per_s": r["value"],
                              "grad_ms_per_step": None if r["grad_step"] is None else round(r["grad_step"]["ms_per_step"], 4),
                              "frac_of_f16_mfma_peak_whole_step": round(r["roofline"]["frac"], 4),
                              "first_forward_ms": round(r["config"]["first_forward_ms"], 2),
                              "open_frac_mean": _mean((r["config"]["uncertified_rows_per_search"] or {}).get("open_frac")),     # steady state:
                              "pair_frac_mean": _mean((r["config"]["uncertified_rows_per_search"] or {}).get("pair_frac")),     # the last timed forward
                              "kmeans_first_forward": r["config"].get("kmeans_first_forward"),
                              "windows_ms_per_step": r["windows_ms_per_step"],
                              "traffic_bytes_per_step": r["roofline"]["traffic"], "traffic_source": r["roofline"]["traffic_source"],
                              "cpu_baseline": r.get("cpu_baseline"), "eager_rocm": r.get("eager_rocm"),
                              "steps": st, "windows": 3, "workload": r["config"]["workload"]}
                except Exception as ex:      # the contract line must not die with an informational one
                    ow[wl] = {"error": f"{type(ex).__name__}: {ex}"}
                torch.cuda.empty_cache()
            out["other_workloads"] = ow
    print(json.dumps(out), flush=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adversarial", action="store_true")
    ap.add_argument("--no-grad-step", action="store_true", help="skip the requires_grad + backward measurement (grad_step)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the compact rvq_cfg3 / grvq_cfg5 / vq_cfg4_shard lines of the default run")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each; the median window is reported")
    ap.add_argument("--workload", default="vq_cfg2", choices=["vq_cfg2", "rvq_cfg3", "grvq_cfg5", "vq_cfg4_shard", "vq_cfg4_sharded"],
                    help="vq_cfg2 (default) is BASELINE.json's headline configuration; the others are informational")
    args = ap.parse_args()
    try:    # the thread that issues the launches keeps its core when other tenants load the shared host (a step is ~15 launches from
        os.setpriority(os.PRIO_PROCESS, 0, -10)     # Python; profiles/r4_final/bench_slow_box.json is what a starved issuer looks like)
        BENCH_ENV["setpriority_minus10"] = True
    except Exception:
        BENCH_ENV["setpriority_minus10"] = False

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the N ranks ourselves (one process per GPU, RCCL), same contract as the driver's command
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
        sys.exit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong GPU count", file=sys.stderr)
        sys.exit(2)
    if args.workload in ("rvq_cfg3", "grvq_cfg5", "vq_cfg4_shard") and world != 1:
        print(f"bench.py: workload {args.workload} is a single-GPU workload", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    try:
        if args.workload == "vq_cfg2":
            vq_cfg2(args, world, rank, dev)
        else:
            line = other_workload(args, world, rank, dev)
            if line is not None:
                print(json.dumps(line), flush=True)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
