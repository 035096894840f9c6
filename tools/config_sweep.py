"""Dev tool: one training step (forward, and forward + backward with an input that requires grad) of VectorQuantize under the options
people combine it with, at cfg-2 size (2^20 rows, D = 256, C = 1024) -- to find the options whose glue costs more than the search.
    python tools/config_sweep.py [rows_log2]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vector_quantize_pytorch_amd import VectorQuantize, ResidualVQ, SimVQ

RVQ_SWEEP = "--rvq" in sys.argv            # the residual modules' options at cfg-3 size (2^18 rows, 8 stages) instead
args = [a for a in sys.argv[1:] if a != "--rvq"]
LOG2 = int(args[0]) if args else (18 if RVQ_SWEEP else 20)
B, T = 64, (1 << LOG2) // 64
dev = torch.device("cuda:0")
CASES = [
    ("plain bf16", dict(dim=256, codebook_size=1024), torch.bfloat16, {}),
    ("plain fp32", dict(dim=256, codebook_size=1024), torch.float32, {}),
    ("fp16 rows", dict(dim=256, codebook_size=1024), torch.float16, {}),
    ("cosine bf16", dict(dim=256, codebook_size=1024, use_cosine_sim=True), torch.bfloat16, {}),
    ("dead-code expiry (threshold 2) bf16", dict(dim=256, codebook_size=1024, threshold_ema_dead_code=2), torch.bfloat16, {}),
    ("dead-code expiry, device-side (expire_without_host_sync) bf16", dict(dim=256, codebook_size=1024, threshold_ema_dead_code=2),
     torch.bfloat16, dict(nosync=True)),
    ("eval bf16", dict(dim=256, codebook_size=1024), torch.bfloat16, dict(eval=True)),
    ("lens mask bf16", dict(dim=256, codebook_size=1024), torch.bfloat16, dict(lens=True)),
    ("codebook_dim 32 (projections) fp32", dict(dim=256, codebook_size=1024, codebook_dim=32), torch.float32, {}),
    ("8 heads x 32 shared codebook fp32", dict(dim=256, codebook_size=1024, heads=8, codebook_dim=32), torch.float32, {}),
    ("8 heads x 32 separate codebooks fp32", dict(dim=256, codebook_size=1024, heads=8, codebook_dim=32, separate_codebook_per_head=True),
     torch.float32, {}),
    ("learnable codebook, no EMA fp32", dict(dim=256, codebook_size=1024, learnable_codebook=True, ema_update=False), torch.float32, {}),
    ("orthogonal reg fp32", dict(dim=256, codebook_size=1024, orthogonal_reg_weight=10.), torch.float32, {}),
    ("rotation trick off (STE) bf16", dict(dim=256, codebook_size=1024, rotation_trick=False), torch.bfloat16, {}),
    ("image fmap fp32", dict(dim=256, codebook_size=1024, accept_image_fmap=True), torch.float32, dict(fmap=True)),
    ("channel_last=False fp32", dict(dim=256, codebook_size=1024, channel_last=False), torch.float32, dict(chan_first=True)),
    ("ResidualVQ 4 stages on a feature map fp32", dict(dim=256, codebook_size=1024, num_quantizers=4, accept_image_fmap=True), torch.float32,
     dict(rvq=True, fmap=True)),
    ("ResidualVQ 4 stages channel-last fp32", dict(dim=256, codebook_size=1024, num_quantizers=4), torch.float32, dict(rvq=True)),
    ("SimVQ fp32", dict(dim=256, codebook_size=1024), torch.float32, dict(sim=True)),
    ("SimVQ bf16 rows", dict(dim=256, codebook_size=1024), torch.bfloat16, dict(sim=True)),
    ("commitment_weight 0 bf16", dict(dim=256, codebook_size=1024, commitment_weight=0.), torch.bfloat16, {}),
    ("ResidualVQ 4 stages, quantize_dropout fp32", dict(dim=256, codebook_size=1024, num_quantizers=4, quantize_dropout=True), torch.float32,
     dict(rvq=True)),
]
if RVQ_SWEEP:
    R = dict(dim=256, num_quantizers=8, codebook_size=1024)
    CASES = [
        ("shared codebook fp32 (cfg 3)", dict(R, shared_codebook=True), torch.float32, dict(rvq=True)),
        ("separate codebooks fp32", dict(R), torch.float32, dict(rvq=True)),
        ("separate codebooks bf16", dict(R), torch.bfloat16, dict(rvq=True)),
        ("dead-code expiry (threshold 2) fp32", dict(R, threshold_ema_dead_code=2), torch.float32, dict(rvq=True)),
        ("dead-code expiry, device-side (expire_without_host_sync) fp32", dict(R, threshold_ema_dead_code=2), torch.float32,
         dict(rvq=True, nosync=True)),
        ("shared codebook + dead-code expiry fp32", dict(R, shared_codebook=True, threshold_ema_dead_code=2), torch.float32, dict(rvq=True)),
        ("cosine fp32", dict(R, use_cosine_sim=True), torch.float32, dict(rvq=True)),
        ("cosine + expiry fp32", dict(R, use_cosine_sim=True, threshold_ema_dead_code=2), torch.float32, dict(rvq=True)),
        ("quantize_dropout fp32", dict(R, quantize_dropout=True, quantize_dropout_cutoff_index=2), torch.float32, dict(rvq=True)),
        ("rotation trick off (STE) fp32", dict(R, rotation_trick=False), torch.float32, dict(rvq=True)),
        ("codebook_dim 64 (projections) fp32", dict(R, codebook_dim=64), torch.float32, dict(rvq=True)),
        ("lens mask fp32", dict(R), torch.float32, dict(rvq=True, mask=True)),
        ("eval fp32", dict(R), torch.float32, dict(rvq=True, eval=True)),
        ("eval, return_all_codes fp32", dict(R), torch.float32, dict(rvq=True, eval=True, all_codes=True)),
        ("dim 512, 4 stages fp32", dict(dim=512, num_quantizers=4, codebook_size=1024), torch.float32, dict(rvq=True, dim=512)),
        ("feature map fp32", dict(R, accept_image_fmap=True), torch.float32, dict(rvq=True, fmap=True)),
    ]


def tm(fn, n=8):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


for name, kw, dt, opt in CASES:
    try:
        torch.manual_seed(0)
        mod = (SimVQ(**kw) if opt.get("sim") else ResidualVQ(**kw) if opt.get("rvq") else VectorQuantize(**kw)).to(dev)
        mod = mod.eval() if opt.get("eval") else mod.train()
        if opt.get("nosync"):
            for m in mod.modules():
                if hasattr(m, "expire_without_host_sync"):
                    m.expire_without_host_sync = True
        dim = opt.get("dim", 256)
        if opt.get("fmap"):
            shape = (B, 256, 64, T // 64)
        elif opt.get("chan_first"):
            shape = (B, 256, T)
        else:
            shape = (B, T, dim)
        x = torch.randn(*shape, device=dev).to(dt)
        call_kw = {}
        if opt.get("mask"):
            call_kw["mask"] = torch.arange(T, device=dev)[None, :] < torch.randint(T // 2, T + 1, (B,), device=dev)[:, None]
        if opt.get("all_codes"):
            call_kw["return_all_codes"] = True
        if opt.get("lens"):
            call_kw["lens"] = torch.randint(T // 2, T + 1, (B,), device=dev)
        res = {}
        for grad in (False, True):
            xi = x.clone().requires_grad_(grad)
            gq = torch.randn_like(x)

            def step():
                out = mod(xi, **call_kw)
                if grad:
                    torch.autograd.backward((out[0], out[2].sum()), (gq, None))
                    xi.grad = None
            with torch.set_grad_enabled(grad):
                res[grad] = tm(step)
            if opt.get("eval"):
                break
        print(f"{name:64s} forward {res[False]:7.3f} ms" + (f"   forward + backward {res[True]:7.3f} ms" if True in res else ""), flush=True)
    except Exception as e:
        print(f"{name:64s} FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
    del mod, x
    torch.cuda.empty_cache()
