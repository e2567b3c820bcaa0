#!/usr/bin/env python3
"""Two engine contexts of ONE process on two HIP streams, driven by two host threads at the same time (include/sgmse_hip.h: one context per
(device, stream), not re-entrant per context): every result against the context's own solo result, bit for bit."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from parity import NET_CASES, make_backbone, make_model
from oracle import synth
from sgmse_amd import _lib
_lib.load_library()
dev = "cuda"
g = torch.Generator().manual_seed(3)
jobs = []
for i, (name, T, B) in enumerate((("fwd_nf128", 64, 1), ("fwd_nf128", 128, 2))):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        net = make_backbone(NET_CASES[name], dev)[0]
        x = (torch.randn(B, 2, 256, T, dtype=torch.complex64, generator=g) * 0.3).to(dev); t = torch.full((B,), 0.3 + 0.2 * i).to(dev)
        ref = net(x, t).cpu()
    jobs.append((st, net, x, t, ref))
res = [None, None]
def work(i):
    st, net, x, t, ref = jobs[i]
    ok = 0
    with torch.cuda.stream(st):
        for _ in range(25):
            ok += int(torch.equal(net(x, t).cpu(), ref))
    res[i] = ok
th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
print("two contexts, two streams, two host threads: forwards identical to the solo results:", res, "of 25 each")
