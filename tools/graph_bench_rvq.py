"""Dev tool: ResidualVQ cfg 3 / GroupedResidualVQ cfg 5 train step, eager vs replayed HIP graph -- how much of the eager step is
host-side launch work (the modules issue ~15 launches per stage from Python); and the same for small batches (8 192 rows, the regime
of a codec training step), with dead-code replacement (device-side under capture).
    python tools/graph_bench_rvq.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import GroupedResidualVQ, ResidualVQ
dev = torch.device("cuda:0")
def tm(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, mk, shape in (("rvq_cfg3", lambda: ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True), (32, 8192, 256)),
                        ("grvq_cfg5", lambda: GroupedResidualVQ(dim=512, groups=4, num_quantizers=8, codebook_size=4096, kmeans_init=True), (32, 8192, 512)),
                        ("rvq 8 stages, 8192 rows", lambda: ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024), (8, 1024, 256)),
                        ("rvq 8 stages, 8192 rows, threshold_ema_dead_code=2",
                         lambda: ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, threshold_ema_dead_code=2), (8, 1024, 256)),
                        ("grvq 2 groups x 8 stages, 8192 rows, threshold_ema_dead_code=2",
                         lambda: GroupedResidualVQ(dim=256, groups=2, num_quantizers=8, codebook_size=1024, threshold_ema_dead_code=2), (8, 1024, 256))):
    torch.manual_seed(0)
    m = mk().to(dev).train()
    x = torch.randn(*shape, device=dev)
    with torch.no_grad():
        m(x); m(x)
        t_eager = tm(lambda: m(x))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(x)
        t_graph = tm(g.replay)
    print(f"{name}: eager {t_eager:.3f} ms, graph replay {t_graph:.3f} ms", flush=True)
