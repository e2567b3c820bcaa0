#!/usr/bin/env python3
"""Side stream on / off: same bits?  (full-width forward, B = 1, T = 64 and 512, repeated)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from parity import NET_CASES, make_backbone
from sgmse_amd import _lib
_lib.load_library()
g = torch.Generator().manual_seed(3)
for T in (64, 512):
    x = (torch.randn(1, 2, 256, T, dtype=torch.complex64, generator=g) * 0.3).cuda(); t = torch.tensor([0.4]).cuda()
    os.environ["SGMSE_SIDE_STREAM"] = "0"
    ref = make_backbone(NET_CASES["fwd_nf128"], "cuda")[0](x, t).cpu()
    os.environ["SGMSE_SIDE_STREAM"] = "1"
    net = make_backbone(NET_CASES["fwd_nf128"], "cuda")[0]
    print(f"T = {T}: side stream on, forwards identical to the one-stream result:", [bool(torch.equal(net(x, t).cpu(), ref)) for _ in range(12)], flush=True)
