#!/usr/bin/env python3
"""Which launch of the forward reads LDS it never wrote?  (SGMSE_POISON_LDS: NaN patterns in every CU's LDS in front of a launch)
Compares the eager forward with and without the poison, then poisons one launch at a time and lists the launches that change the result."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from parity import NET_CASES, make_backbone
from sgmse_amd import _lib

_lib.load_library()
name = os.environ.get("PROBE_NET", "fwd_nf128")
T = int(os.environ.get("PROBE_T", "64"))
g = torch.Generator().manual_seed(3)
x = (torch.randn(1, 2, 256, T, dtype=torch.complex64, generator=g) * 0.3).cuda()
t = torch.tensor([0.4]).cuda()


def run(env):
    for k, v in env.items():
        os.environ[k] = v
    net = make_backbone(NET_CASES[name], "cuda")[0]
    o = net(x, t).cpu()
    for k in env:
        os.environ.pop(k, None)
    return o


ref = run({})
all_ = run({"SGMSE_POISON_LDS": "1"})
print("poison in front of every launch: identical", bool(torch.equal(ref, all_)), "finite", bool(torch.isfinite(torch.view_as_real(all_)).all()),
      "max |diff| %.2e" % float((all_ - ref).abs().nan_to_num(1e9).max()), flush=True)
if not torch.equal(ref, all_):
    n = int(os.environ.get("PROBE_LAUNCHES", "300"))
    bad = []
    os.environ["SGMSE_DEBUG_SYNC"] = "0"
    for i in range(n):
        o = run({"SGMSE_POISON_LDS": "1", "SGMSE_POISON_LDS_AT": str(i)})
        if not torch.equal(o, ref):
            bad.append(i)
    print("launch indices (tock order) whose poisoned LDS changes the result:", bad, flush=True)
