#!/usr/bin/env python3
"""Op by op: does a kernel's result depend on what else runs on the GPU?  Every op is run once alone (reference) and then REPS times
while a second process loads the device (tools/probes/concurrency_bits_probe.py --load)."""
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sgmse_amd import _lib, ops

_lib.load_library()
dev = "cuda"
g = torch.Generator().manual_seed(1)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
cases = {}


def conv_case(name, B, Ci, Co, H, W, ks=3, **kw):
    x, w, b, r = R(B, Ci, H, W), R(Co, Ci, ks, ks) / math.sqrt(Ci * ks * ks), R(Co), R(B, Co, H, W)
    sc, sh = R(B, Ci), R(B, Ci)
    xf = kw.pop("xform", True)
    cases[name] = lambda: ops.conv2d(x, w, b, residual=r, out_scale=0.7, in_scale=sc if xf else None, in_shift=sh if xf else None, in_act=xf, **kw)


conv_case("conv3x3 fp32 mfma 128->128 @128x32", 1, 128, 128, 128, 32)
conv_case("conv3x3 fp32 mfma 256->256 @16x4", 1, 256, 256, 16, 4)
conv_case("conv3x3 fp32 mfma 256->256 @32x8", 1, 256, 256, 32, 8)
conv_case("conv3x3 fp32 mfma 8->128 @256x64 (entry)", 1, 8, 128, 256, 64, xform=False)
conv_case("conv3x3 wino 128->128 @256x64", 1, 128, 128, 256, 64, force_split="wino")
conv_case("conv3x3 wino4 128->128 @128x32", 1, 128, 128, 128, 32, force_split="wino4")
conv_case("conv3x3 fp16x2 256->256 @32x8", 1, 256, 256, 32, 8, force_split="fp16x2")
conv_case("conv3x3 thin 128->4 @256x64", 1, 128, 4, 256, 64, force_split="thin")
if "--more" in sys.argv or "--as-load" in sys.argv:          # further candidates for the aggressor side (find_aggressor.sh MORE=1)
    conv_case("conv3x3 bf16x3 256->256 @32x8", 1, 256, 256, 32, 8, force_split="bf16x3")
    conv_case("conv3x3 fp16x2 128->4 @64x16 (thin split shape)", 1, 128, 4, 64, 16, force_split="fp16x2")
    conv_case("conv1x1 fp16x2 256->128 @64x16", 1, 256, 128, 64, 16, ks=1, xform=False, force_split="fp16x2")
    conv_case("conv3x3 fp16x2 128->128 @256x64", 1, 128, 128, 256, 64, force_split="fp16x2")
conv_case("conv1x1 fp32 256->128 @64x16", 1, 256, 128, 64, 16, ks=1, xform=False)
conv_case("conv1x1 direct 4->128 @128x32", 1, 4, 128, 128, 32, ks=1, xform=False, force_direct=True)
xg, wg, bg = R(1, 128, 128, 32), R(128), R(128)
cases["group_norm 128 @128x32"] = lambda: ops.group_norm(xg, wg, bg, act=True)
xg2 = R(1, 256, 16, 4)
wg2, bg2 = R(256), R(256)
cases["group_norm 256 @16x4"] = lambda: ops.group_norm(xg2, wg2, bg2, act=True)
xf_, scf, shf = R(1, 128, 128, 32), R(1, 128), R(1, 128)
cases["fir up tiled 128 @128x32 +producer +raw"] = lambda: torch.cat([t.flatten() for t in ops.fir_resample(xf_, True, in_scale=scf, in_shift=shf, in_act=True, return_raw=True)])
cases["fir down tiled 128 @128x32 +producer +raw"] = lambda: torch.cat([t.flatten() for t in ops.fir_resample(xf_, False, in_scale=scf, in_shift=shf, in_act=True, return_raw=True)])
x4 = R(1, 4, 64, 16)
cases["fir up 4 @64x16"] = lambda: ops.fir_resample(x4, True)
cases["fir down 4 @64x16"] = lambda: ops.fir_resample(x4, False)
qkv = R(1, 768, 64)
cases["attention C=256 S=64"] = lambda: ops.attention(qkv)
qkv2 = R(1, 768, 512)
cases["attention C=256 S=512"] = lambda: ops.attention(qkv2)

if "--as-load" in sys.argv:
    # run ONE op in a loop (the aggressor side of tools/probes/pk_fma_probe.hip): which kernel disturbs packed FMAs of another process?
    name = sys.argv[sys.argv.index("--as-load") + 1]
    f = cases[name]
    t0 = time.time()
    while time.time() - t0 < float(os.environ.get("LOAD_SECONDS", "8")):
        for _ in range(50):
            f()
        torch.cuda.synchronize()
    sys.exit(0)
if "--list" in sys.argv:
    print("\n".join(cases))
    sys.exit(0)
ref = {k: f().cpu() for k, f in cases.items()}
for k, f in cases.items():
    assert torch.equal(f().cpu(), ref[k]), k + " is not reproducible even alone"
print("alone: every op reproducible", flush=True)
load = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "probes", "concurrency_bits_probe.py"), "--load"],
                        env=dict(os.environ, LOAD_SECONDS=os.environ.get("LOAD_SECONDS", "60")))
time.sleep(12)
n = int(os.environ.get("REPS", "60"))
for k, f in cases.items():
    bad, worst = 0, 0.0
    for _ in range(n):
        o = f().cpu()
        if not torch.equal(o, ref[k]):
            bad += 1
            worst = max(worst, float((o - ref[k]).abs().nan_to_num(1e9).max()))
    print(f"{k:52s} {n - bad:3d} of {n} identical" + (f"   worst |diff| {worst:.2e}" if bad else ""), flush=True)
load.wait()
