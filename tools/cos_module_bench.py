import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vector_quantize_pytorch_amd import VectorQuantize
for dt in (torch.bfloat16, torch.float32):
    for scr in ("1", "0"):
        os.environ["VQHIP_SCREEN"] = scr
        torch.manual_seed(0)
        vq = VectorQuantize(dim=256, codebook_size=1024, use_cosine_sim=True).cuda().train()
        x = torch.randn(64, 16384, 256, device="cuda").to(dt)
        with torch.no_grad():
            for _ in range(3): out = vq(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): out = vq(x)
            torch.cuda.synchronize()
        print(f"cosine VQ train step {dt} screen={scr}: {(time.perf_counter()-t0)/10*1e3:.3f} ms  loss {out[2].item():.6f} idxsum {out[1].sum().item()}", flush=True)
