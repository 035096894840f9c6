"""Dev tool: host enqueue time vs GPU time of one training forward (is a step launch-bound?).  Enqueue time = wall time of K forwards
WITHOUT a final synchronize (the queue never fills at these sizes); step time = the same with one synchronize at the end.
    python tools/host_overhead.py [rvq_cfg3|grvq_cfg5|vq_cfg2]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import GroupedResidualVQ, ResidualVQ, VectorQuantize
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "rvq_cfg3"
torch.manual_seed(0)
if wl == "rvq_cfg3":
    m, x = ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True), torch.randn(32, 8192, 256, device=dev)
elif wl == "grvq_cfg5":
    m, x = GroupedResidualVQ(dim=512, groups=4, num_quantizers=8, codebook_size=4096, kmeans_init=True), torch.randn(32, 8192, 512, device=dev)
else:
    m, x = VectorQuantize(dim=256, codebook_size=1024), torch.randn(64, 16384, 256, device=dev).bfloat16()
m = m.to(dev).train()
K = 20
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        m(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"{wl} chunks={os.environ.get('VQHIP_RVQ_CHUNKS', 'default')}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms/step, step {1e3 * (t2 - t0) / K:.3f} ms")
