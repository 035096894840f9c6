"""Runs golden fixtures by name and prints, per step, where this package differs from the reference (index mismatches with their
positions, loss, quantized, gradients, state after): the triage view behind tests/test_gpu_modules.py::test_module_matches_reference_golden.

    python tools/combo_debug.py combo_004 combo_139 ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import golden_util as G
import test_gpu_modules as T

dev = torch.device("cuda", 0)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if a.shape != b.shape:
        return f"SHAPE {tuple(a.shape)} vs {tuple(b.shape)}"
    return f"{(a - b).abs().max().item() / max(b.abs().max().item(), 1e-12):.2e}"


for name in sys.argv[1:]:
    fx = G.Fixture(name)
    print(f"== {name} {fx.meta['cls']} {fx.meta['kwargs']} train={fx.meta['train']} grad={fx.meta['grad']} fwd={list(fx.meta['fwd_kwargs'])}")
    try:
        mod = T._build(fx, dev)
        for s in range(fx.meta["steps"]):
            x = fx.t(f"x{s}").to(dev)
            if fx.meta["grad"]:
                x.requires_grad_(True)
            res = mod(x, **fx.fwd_kwargs(dev))
            if len(res) == 2:
                res = (res[0], torch.zeros(1, dtype=torch.long, device=dev), res[1])
            q, idx, loss = res[:3]
            want = fx.t(f"idx{s}")
            bad = (idx.cpu() != want).nonzero()
            print(f"  step {s}: idx mismatches {bad.shape[0]} of {want.numel()} at {bad[:6].tolist()}  "
                  f"mine {idx.cpu()[tuple(bad[:6].T)].tolist() if bad.numel() else []} want {want[tuple(bad[:6].T)].tolist() if bad.numel() else []}")
            print(f"          loss mine {loss.reshape(-1)[:8].tolist()} want {fx.t(f'loss{s}').reshape(-1)[:8].tolist()}")
            if fx.has(f"q{s}"):
                print(f"          q rel err {rel(q, fx.t(f'q{s}'))}")
            if fx.meta["grad"] or fx.meta.get("param_grad"):
                for p_ in mod.parameters():
                    p_.grad = None
                (loss.sum() * 3.0 + (q * fx.t(f"gw{s}").to(dev)).sum()).backward()
                if fx.meta["grad"]:
                    print(f"          grad_x rel err {rel(x.grad, fx.t(f'gx{s}'))}")
                for k in fx.arr:
                    if k.startswith(f"pg{s}/"):
                        n = k[len(f"pg{s}/"):]
                        g = dict(mod.named_parameters())[n].grad
                        print(f"          grad {n}: {'MISSING' if g is None else rel(g, fx.t(k))}")
        if fx.meta["train"]:
            after, mine = fx.state("after"), mod.state_dict()
            for k, v in after.items():
                if not k.endswith("initted"):
                    e = rel(mine[k].float(), v.float())
                    if not e[0].isdigit() or float(e) > 1e-5:
                        print(f"  after {k}: rel err {e}")
    except Exception as e:
        import traceback
        traceback.print_exc(limit=-4)
