"""Dev tool: training step of a cosine-similarity VectorQuantize (use_cosine_sim=True, cfg-2 sizes and a small batch) through the
fused train step (default), the separate calls with the loss from the statistics pass, and the separate calls with the loss summed
by the search kernel; with an input that requires grad also the round-3 path, F.normalize as autograd ops.
    python tools/cos_module_bench.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vector_quantize_pytorch_amd import VectorQuantize

MODES = [("fused step", dict(VQHIP_FUSED_STEP="1", VQHIP_STATS_SQERR="1", VQHIP_L2NORM_FN="1")),
         ("separate calls, loss in the statistics pass", dict(VQHIP_FUSED_STEP="0", VQHIP_STATS_SQERR="1", VQHIP_L2NORM_FN="1")),
         ("separate calls, loss in the search kernel", dict(VQHIP_FUSED_STEP="0", VQHIP_STATS_SQERR="0", VQHIP_L2NORM_FN="1")),
         ("separate calls, loss in the search kernel, l2norm as F.normalize's autograd ops (round 3)",
          dict(VQHIP_FUSED_STEP="0", VQHIP_STATS_SQERR="0", VQHIP_L2NORM_FN="0"))]
for dt in (torch.bfloat16, torch.float32):
    for shape in ((64, 16384, 256), (8, 1024, 256)):
        for grad in (False, True):
            for name, env in MODES:
                if not grad and env["VQHIP_L2NORM_FN"] == "0":
                    continue
                os.environ.update(env)
                torch.manual_seed(0)
                vq = VectorQuantize(dim=256, codebook_size=1024, use_cosine_sim=True).cuda().train()
                x = torch.randn(*shape, device="cuda").to(dt).requires_grad_(grad)
                gq = torch.randn_like(x)

                def step():
                    out = vq(x)
                    if grad:
                        torch.autograd.backward((out[0], out[2].sum()), (gq, None))
                        x.grad = None
                    return out
                with torch.set_grad_enabled(grad):
                    for _ in range(3): out = step()
                    best = 1e9
                    for w in range(3):                       # (min over three windows: allocator / first-launch hiccups excluded)
                        torch.cuda.synchronize(); t0 = time.perf_counter()
                        for _ in range(10): out = step()
                        torch.cuda.synchronize()
                        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
                print(f"cosine VQ train step {str(dt)[6:]} {shape} grad={int(grad)} {name}: {best:.3f} ms  "
                      f"loss {out[2].item():.6f} idxsum {out[1].sum().item()}", flush=True)
