"""Dev tool: VectorQuantize train step, eager vs replayed HIP graph, small / medium N (launch-bound regime)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import VectorQuantize
dev = torch.device("cuda:0")
def tm(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n in (1024, 16384, 65536, 262144):
    torch.manual_seed(0)
    vq = VectorQuantize(dim=256, codebook_size=1024).to(dev).train()
    x = torch.randn(1, n, 256, device=dev).bfloat16()
    with torch.no_grad():
        t_eager = tm(lambda: vq(x))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = vq(x)
        t_graph = tm(g.replay)
    print(f"rows {n}: eager {t_eager:.0f} us, graph replay {t_graph:.0f} us", flush=True)
