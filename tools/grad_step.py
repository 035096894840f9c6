"""Dev tool: a few training steps with an input that requires grad (forward + backward), for rocprofv3 --kernel-trace --stats.
    python tools/grad_step.py [vq_cfg2|vq_cos|rvq_cfg3|grvq_cfg5] [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_quantize_pytorch_amd import GroupedResidualVQ, ResidualVQ, VectorQuantize
dev = torch.device("cuda:0")
wl = sys.argv[1] if len(sys.argv) > 1 else "rvq_cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
if wl == "vq_cfg2":
    mod, shape, dt = VectorQuantize(dim=256, codebook_size=1024), (64, 16384, 256), torch.bfloat16
elif wl == "vq_cos":
    mod, shape, dt = VectorQuantize(dim=256, codebook_size=1024, use_cosine_sim=True), (64, 16384, 256), torch.bfloat16
elif wl == "rvq_cfg3":
    mod, shape, dt = ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True), (32, 8192, 256), torch.float32
else:
    mod, shape, dt = GroupedResidualVQ(dim=512, groups=4, num_quantizers=8, codebook_size=4096, kmeans_init=True), (32, 8192, 512), torch.float32
mod = mod.to(dev).train()
x = torch.randn(*shape, device=dev).to(dt).requires_grad_(True)
gq = torch.randn_like(x)
for i in range(steps + 2):
    x.grad = None
    res = mod(x)
    torch.autograd.backward((res[0], res[2].sum()), (gq, None))
torch.cuda.synchronize()
print("done", wl)
