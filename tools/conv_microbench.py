#!/usr/bin/env python3
"""A/B of the MFMA convolution kernel variants on the dominant shapes (random operands, HIP-event timing of
back-to-back launches; interleaved rounds, median reported)."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgmse_amd import _lib

VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "4,3076,5124,9220,17412").split(",")]
SHAPES = [  # ks, B, Cin, Cout, H, W
    (3, 8, 128, 128, 256, 512),
    (3, 8, 256, 128, 256, 512),
    (3, 16, 256, 256, 64, 128),
    (1, 8, 256, 128, 256, 512),
]
ROUNDS = int(os.environ.get("ROUNDS", "3"))
_lib.load_library()
ctx = _lib.Context("cuda")
res = {}
for (ks, B, ci, co, H, W) in SHAPES:
    flops = 2.0 * B * co * ci * ks * ks * H * W
    for fused in (0, 1):
        times = {v: [] for v in VARIANTS}
        for r in range(ROUNDS):
            for v in VARIANTS:
                times[v].append(ctx.bench_conv(ks, B, ci, co, H, W, variant=v, iters=5, fused=bool(fused)))
        for v in VARIANTS:
            med = statistics.median(times[v])
            res[f"ks{ks}_B{B}_{ci}to{co}_{H}x{W}_fused{fused}_v{v}"] = {"ms": round(med, 4), "tflops": round(flops / med / 1e9, 1),
                                                                      "min_ms": round(min(times[v]), 4)}
            print(f"ks={ks} B={B} {ci}->{co} {H}x{W} fused={fused} variant={v}: {med:8.3f} ms  {flops / med / 1e9:7.1f} TFLOP/s", flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "conv_microbench.json"), "w"), indent=1)
