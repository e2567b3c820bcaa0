#!/usr/bin/env python3
"""Board power and shader clock (rocm-smi) while ONE kernel variant runs back to back: is the dominant kernel power-limited?
Usage: power_probe.py  (variants: fp16x2 full kernel, MFMA-only loop without epilogue memory traffic, fp32 MFMA kernel)"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sgmse_amd import _lib

_lib.load_library(os.environ.get("SGMSE_LIB_PATH"))      # (measurement: an alternative build of the library)
ctx = _lib.Context("cuda")


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            p = next((v for k, v in card.items() if "Power" in k and "W" in k), None)
            sclk = next((v for k, v in card.items() if k.lower().startswith("sclk clock speed")), None)
            out.append((time.time(), p, sclk))
        except Exception as e:      # noqa: BLE001
            out.append((time.time(), None, str(e)[:60]))
        time.sleep(0.15)


CASES = {"split": ("fp16x2 split kernel (full)", 128), "mfma_only": ("fp16x2 MFMA-only loop, no epilogue memory (ablation 59)", 128 + (59 << 12)),
         "no_staging": ("fp16x2 no staging (ablation 8)", 128 + (8 << 12)), "fp32": ("fp32 MFMA kernel", 0),
         "wino": ("Winograd F(2,3) x fp16x2 kernel", 128 + 1024),
         "wino2d": ("2-D Winograd F(2x2,3x3) x fp16x2 kernel", 128 + 2048),
         "wino2d_mfma_only": ("2-D Winograd kernel without staging behind the prologue (ablation 8)", 128 + 2048 + (8 << 12)),
         "wino_mfma_only": ("Winograd kernel without staging and raw loads behind the prologue (ablation 24; ABLATION=1 build)", 128 + 1024 + (24 << 12))}
ITERS = int(os.environ.get("ITERS", "3500"))


def sampled(fn):
    stop, log = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, log)); th.start()
    t0 = time.time()
    r = fn()
    torch.cuda.synchronize()
    dt = time.time() - t0
    stop.set(); th.join()
    mid = [(float(p), c) for (t, p, c) in log if t0 + 0.4 < t < t0 + dt - 0.1 and p is not None]
    return r, dt, mid


if os.environ.get("STATIC") == "1":
    # The static term of the energy model: socket power with nothing running, and under an HBM copy stream at
    # full and at ~half duty (the same kernel, the same clocks; copy bursts alternating with equally long sleeps) -- two points on
    # P = P_static + e_hbm * bytes/s, solved for both unknowns.
    x = torch.empty(1 << 29, dtype=torch.float32, device="cuda").normal_()      # 2 GiB
    y = torch.empty_like(x)
    _, _, idle = sampled(lambda: time.sleep(2.5))
    pw = lambda m: sum(p for p, _ in m) / max(len(m), 1)
    print(f"idle: {pw(idle):.0f} W  {idle[:4]}", flush=True)
    res = {}
    for name, duty in (("copy_full", 1.0), ("copy_half", 0.5)):
        y.copy_(x); torch.cuda.synchronize()
        t0 = time.time(); y.copy_(x); torch.cuda.synchronize(); t_copy = time.time() - t0
        n = int(2.5 / (t_copy / duty))
        def run():
            for _ in range(n):
                y.copy_(x)
                if duty < 1.0:
                    torch.cuda.synchronize(); time.sleep(t_copy * (1.0 / duty - 1.0))
        _, dt, m = sampled(run)
        bw = n * 2 * x.numel() * 4 / dt / 1e12
        res[name] = (pw(m), bw)
        print(f"{name}: {pw(m):.0f} W at {bw:.2f} TB/s of HBM traffic (read + write), {m[:3]}", flush=True)
    (p1, b1), (p2, b2) = res["copy_full"], res["copy_half"]
    e_hbm = (p1 - p2) / max(b1 - b2, 1e-9)          # W per TB/s = pJ per byte
    print(f"two-point fit: e_hbm = {e_hbm:.0f} pJ/B, P_static (memory clocks up, shader idle) = {p1 - e_hbm * b1:.0f} W; idle {pw(idle):.0f} W", flush=True)
    del x, y

for name, variant in (CASES[k] for k in os.environ.get("CASES", "split,wino,fp32").split(",") if k):
    ctx.bench_conv(3, 8, 128, 128, 256, 512, variant=variant, iters=20, fused=True)      # warm
    ms, dt, mid = sampled(lambda: ctx.bench_conv(3, 8, 128, 128, 256, 512, variant=variant, iters=ITERS, fused=True))
    mean = sum(p for p, _ in mid) / max(len(mid), 1)
    print(f"{name}: {ms:.3f} ms per launch over {dt:.1f} s; mean {mean:.0f} W, {ms * mean * 1e-3:.3f} J per launch; samples (power W, sclk): {mid[:6]}", flush=True)
