"""The fused train step with the EMA fold inside the segmented sum (VQHIP_STEP_FOLD=1: the wave that adds a code's last chunk folds the
code's row, the workgroup that finishes last reduces the loss) against the default (vq_step_fold_kernel, a launch of its own): many
steps on one reused workspace (the tickets are re-armed by every step's scan kernel), empty codes, one-code batches, bf16 and fp32
rows.  Two child processes (the switch is read once per process); indices and cluster sizes must be identical, embed / embed_avg / loss
equal up to the order of the fp32 atomics.   python tools/fold_stress.py [--quick]"""
import os, sys, subprocess, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL = [(60, 1 << 20, 1024, 256, torch.bfloat16), (60, 200000, 512, 64, torch.float32), (40, 70001, 300, 128, torch.float32),
        (40, 1 << 18, 4096, 32, torch.bfloat16)]
QUICK = [(10, 1 << 18, 1024, 256, torch.bfloat16), (12, 200000, 512, 64, torch.float32), (10, 70001, 300, 128, torch.float32),
         (8, 40000, 4096, 32, torch.bfloat16)]


def run(steps, N, C, D, dtype):
    from vector_quantize_pytorch_amd import _lib as L
    dev = "cuda"
    embed = torch.empty(C, D, device=dev); embed_avg = torch.empty(C, D, device=dev); cs = torch.empty(C, device=dev)
    outs = []
    g = torch.Generator(device=dev); g.manual_seed(1)
    for s in range(steps):
        # every step starts from its own seeded state (the sums are accumulated with atomics: two runs of ONE mode already differ in the
        # last bits of embed, and a drifting codebook would move near-tie rows); the workspace and its tickets are reused step after step
        embed.copy_(torch.randn(C, D, device=dev, generator=g)); embed_avg.copy_(embed * 1.25); cs.fill_(1.5)
        x = torch.randn(N, D, device=dev, generator=g).to(dtype)
        if s % 7 == 3: x[:, :] = x[:1, :]          # every row to one code: C - 1 empty codes, one long segment
        r = L.vq_train_step(x, embed, embed_avg, cs, decay=0.8, eps=1e-5, loss_scale=1.0 / x.numel(), fold=True, reuse_scratch=True)
        outs.append((embed.clone(), embed_avg.clone(), cs.clone(), r["loss"].clone(), r["idx"].clone()))
    torch.cuda.synchronize()
    return outs


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        mode, quick = sys.argv[2], sys.argv[3] == "1"
        os.environ["VQHIP_STEP_FOLD"] = mode
        torch.save([run(*c) for c in (QUICK if quick else FULL)], f"/tmp/fold_{mode}.pt")
    else:
        quick = "1" if "--quick" in sys.argv else "0"
        for m in ("0", "1"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", m, quick], cwd=ROOT)
        a, b = torch.load("/tmp/fold_0.pt"), torch.load("/tmp/fold_1.pt")
        worst = 0.0
        for ci, (ra, rb) in enumerate(zip(a, b)):
            for s, (oa, ob) in enumerate(zip(ra, rb)):
                assert torch.equal(oa[4], ob[4]), ("idx", ci, s)
                assert torch.equal(oa[2], ob[2]), ("cluster_size", ci, s)
                for k in (0, 1, 3):
                    d = (oa[k] - ob[k]).abs().max().item(); ref = oa[k].abs().max().item() + 1e-30
                    worst = max(worst, d / ref)
                    assert d / ref < 2e-5, (k, ci, s, d, ref)
        print("fold stress OK: identical indices / cluster sizes, worst relative difference of embed / embed_avg / loss %.3g" % worst)
