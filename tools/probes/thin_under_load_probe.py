#!/usr/bin/env python3
"""conv3x3_thin_kernel under outside load: where do the differing outputs sit?"""
import math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sgmse_amd import _lib, ops
_lib.load_library()
dev = "cuda"
g = torch.Generator().manual_seed(1)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
B, Ci, H, W = 1, 128, 256, 64
x, w, b, r = R(B, Ci, H, W), R(4, Ci, 3, 3) / math.sqrt(Ci * 9), R(4), R(B, 4, H, W)
sc, sh = R(B, Ci), R(B, Ci)
variants = {
    "producer + residual": lambda: ops.conv2d(x, w, b, residual=r, out_scale=0.7, in_scale=sc, in_shift=sh, in_act=True, force_split="thin"),
}
ref = {k: f().cpu() for k, f in variants.items()}
load = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "probes", "concurrency_bits_probe.py"), "--load"], env=dict({k: v for k, v in os.environ.items() if k != "SGMSE_THIN_VARIANT"}, LOAD_SECONDS="25"))
time.sleep(12)
for k, f in variants.items():
    bad = 0
    shown = 0
    for it in range(40):
        o = f().cpu()
        if not torch.equal(o, ref[k]):
            bad += 1
            if shown < 3:
                shown += 1
                d = (o != ref[k])[0]                      # [4, H, W]
                ys, xs = torch.nonzero(d.any(0), as_tuple=True)
                tiles = sorted({(int(y) // 16, int(xx) // 64) for y, xx in zip(ys.tolist(), xs.tolist())})
                per_ch = d.flatten(1).sum(1).tolist()
                rows = sorted({int(y) % 16 for y in ys.tolist()})
                cols = sorted({int(xx) % 64 for xx in xs.tolist()})
                print(f"  [{k}] run {it}: {int(d.sum())} differing values; per channel {per_ch}; tiles (ty,tx) {tiles[:12]}{'...' if len(tiles) > 12 else ''}; "
                      f"rows within tile {rows[:18]}; {len(cols)} distinct columns within tile, first {cols[:8]}; max |diff| {float((o - ref[k]).abs().max()):.2e}", flush=True)
    print(f"{k}: {40 - bad} of 40 identical", flush=True)
load.wait()
