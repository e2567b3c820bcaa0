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
         "wino": ("Winograd F(2,3) x fp16x2 kernel", 128 + 1024)}
ITERS = int(os.environ.get("ITERS", "3500"))
for name, variant in (CASES[k] for k in os.environ.get("CASES", "split,mfma_only,no_staging,fp32").split(",")):
    ctx.bench_conv(3, 8, 128, 128, 256, 512, variant=variant, iters=20, fused=True)      # warm
    stop, log = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, log)); th.start()
    t0 = time.time()
    ms = ctx.bench_conv(3, 8, 128, 128, 256, 512, variant=variant, iters=ITERS, fused=True)
    dt = time.time() - t0
    stop.set(); th.join()
    mid = [(p, c) for (t, p, c) in log if t0 + 0.4 < t < t0 + dt - 0.1 and p is not None]
    print(f"{name}: {ms:.3f} ms per launch over {dt:.1f} s; samples (power W, sclk) during the run: {mid[:12]}", flush=True)
