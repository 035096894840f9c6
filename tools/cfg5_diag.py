"""round 6 (VERDICT r5 #1a): why is grvq_cfg5 16.1-17.2 ms inside bench.py's default run and 14.4-15.5 ms standalone, and what flips
windows between 14.8 and 18.7 ms inside one process?

    python tools/cfg5_diag.py <mode> [--windows 10] [--steps 5] [--warmup 2]

modes:  fresh        the module first in the process
        after        after a cfg-2 and a cfg-3 module have run in this process (what bench.py's default line does)
        serial       fresh, GroupedResidualVQ.concurrent_groups = False
Prints one JSON line: per-window ms, and per window the caching allocator's counters (device mallocs / frees, reserved bytes) and
the number of streams in use -- an allocator that still grows its pools inside the timed windows (cross-stream blocks cannot be
reused before their events retire) would show up as `hipMalloc` calls there."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def alloc_counters():
    s = torch.cuda.memory_stats()
    return {"segments_alloc": s.get("segment.all.allocated", 0), "segments_free": s.get("segment.all.freed", 0),
            "reserved_mb": round(s.get("reserved_bytes.all.current", 0) / 2**20, 1),
            "active_mb": round(s.get("active_bytes.all.current", 0) / 2**20, 1),
            "retries": s.get("num_alloc_retries", 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["fresh", "after", "serial", "batched", "streams"])
    ap.add_argument("--windows", type=int, default=10)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--preheat", type=float, default=0.25)
    args = ap.parse_args()
    from vector_quantize_pytorch_amd import GroupedResidualVQ, ResidualVQ, VectorQuantize
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.mode == "after":
        with torch.no_grad():
            vq = VectorQuantize(dim=256, codebook_size=1024).to(dev).train()
            xs = [torch.randn(64, 16384, 256, device=dev).bfloat16() for _ in range(4)]
            for i in range(30):
                vq(xs[i % 4])
            del vq, xs
            torch.cuda.empty_cache()
            r = ResidualVQ(dim=256, num_quantizers=8, codebook_size=1024, shared_codebook=True).to(dev).train()
            xs = [torch.randn(32, 8192, 256, device=dev) for _ in range(2)]
            for i in range(30):
                r(xs[i % 2])
            del r, xs
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    if args.mode == "serial":
        GroupedResidualVQ.concurrent_groups = False
    if args.mode == "batched":
        os.environ["VQHIP_GRVQ_BATCHED"] = "1"
    if args.mode == "streams":
        os.environ["VQHIP_GRVQ_BATCHED"] = "0"
    torch.manual_seed(0)
    gen = torch.Generator(device=dev).manual_seed(1)
    mod = GroupedResidualVQ(dim=512, groups=4, num_quantizers=8, codebook_size=4096, kmeans_init=True).to(dev).train()
    batches = [torch.randn(32, 8192, 512, generator=gen, device=dev) for _ in range(2)]
    out = {"mode": args.mode, "steps": args.steps, "warmup": args.warmup}
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mod(batches[0])
        torch.cuda.synchronize(); out["first_forward_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        out["alloc_after_first"] = alloc_counters()
        for i in range(args.warmup):
            mod(batches[i % 2])
        torch.cuda.synchronize()
        out["alloc_after_warmup"] = alloc_counters()
        bench._preheat(dev, args.preheat)
        wins, allocs = [], []
        k = 0
        for w in range(args.windows):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.steps):
                mod(batches[k % 2]); k += 1
            torch.cuda.synchronize()
            wins.append(round((time.perf_counter() - t0) / args.steps * 1e3, 3))
            a = alloc_counters()
            allocs.append((a["segments_alloc"], a["segments_free"], a["reserved_mb"]))
        out["windows_ms"] = wins
        out["alloc_per_window(segments_alloc, segments_free, reserved_mb)"] = allocs
        # per-step times of 20 more steps (each synchronised): is the bimodality per step or per window?
        per = []
        for _ in range(20):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            mod(batches[k % 2]); k += 1
            torch.cuda.synchronize()
            per.append(round((time.perf_counter() - t0) * 1e3, 2))
        out["single_steps_ms"] = per
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
