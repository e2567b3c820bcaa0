#!/usr/bin/env python3
"""Race hunt: the same forward many times in fresh engines must give the same bits (and the right answer).
Usage: determinism_check.py [reps]   (env knobs such as SGMSE_SPLIT_MIN_TILES apply)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import parity as P
from sgmse_amd import _lib

_lib.load_library()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name in ("fwd_nf128", "fwd_nf32"):
    cfg = P.NET_CASES[name]
    z = P.load(name)
    x, t, ref = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["t"]).cuda(), torch.from_numpy(z["out"])
    outs, errs = [], []
    for r in range(reps):
        net, _ = P.make_backbone(cfg, "cuda")          # fresh engine: weight packing + first forward every time
        o = net(x, t).cpu()
        o2 = net(x, t).cpu()
        outs += [o, o2]
        errs.append(P.rel_l2(o, ref))
    same = sum(torch.equal(o, outs[0]) for o in outs)
    print(f"{name}: {same}/{len(outs)} outputs bit-identical to the first; rel_l2 min {min(errs):.3e} max {max(errs):.3e}", flush=True)
    if same != len(outs):
        for i, o in enumerate(outs):
            if not torch.equal(o, outs[0]):
                d = (o - outs[0]).abs()
                print(f"   run {i}: {int((d > 0).sum())} elements differ, max abs diff {float(d.max()):.3e}, rel_l2 vs ref {P.rel_l2(o, ref):.3e}")
