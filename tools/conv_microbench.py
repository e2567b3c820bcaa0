#!/usr/bin/env python3
"""A/B of the MFMA convolution kernel variants on the dominant shapes (random operands, HIP-event timing of
back-to-back launches; interleaved rounds, median reported).

Env: VARIANTS (kernel-variant ids, see kernels_conv.h), SHAPES (indices into SHAPES below), FUSED (0,1), ROUNDS,
CALIB=1 (also run one GroupNorm-statistics pass over a tensor of known size and record its byte count: a float4
read stream used to calibrate FETCH_SIZE when this script runs under rocprofv3 --pmc)."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sgmse_amd import _lib

VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,4").split(",")]
ALL_SHAPES = [  # ks, B, Cin, Cout, H, W
    (3, 8, 128, 128, 256, 512),
    (3, 8, 256, 128, 256, 512),
    (3, 16, 256, 256, 64, 128),
    (1, 8, 256, 128, 256, 512),
    # the same 128->128 layer with the same tile count (4096) on smaller planes, and at the benched batch: what the per-FLOP deficit of
    # the 256 x 512 level against 128 x 256 in the network follows -- plane size or launch length
    (3, 32, 128, 128, 128, 256),
    (3, 32, 128, 128, 256, 512),
    (3, 128, 128, 128, 64, 128),
    (3, 2, 128, 128, 256, 512),
]
SHAPES = [ALL_SHAPES[int(i)] for i in os.environ.get("SHAPES", "0,1,2,3").split(",")]
FUSED = [int(v) for v in os.environ.get("FUSED", "0,1").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", "3"))
_lib.load_library(os.environ.get("SGMSE_LIB_PATH"))      # (measurement: an alternative build of the library)
ctx = _lib.Context("cuda")
res = {}
for (ks, B, ci, co, H, W) in SHAPES:
    flops = 2.0 * B * co * ci * ks * ks * H * W
    for fused in FUSED:
        times = {v: [] for v in VARIANTS}
        for r in range(ROUNDS):
            for v in VARIANTS:
                times[v].append(ctx.bench_conv(ks, B, ci, co, H, W, variant=v, iters=5, fused=bool(fused)))
        for v in VARIANTS:
            med = statistics.median(times[v])
            alg = 4.0 * B * H * W * (ci + co * (2 if fused else 1)) + 4.0 * co * ci * ks * ks
            res[f"ks{ks}_B{B}_{ci}to{co}_{H}x{W}_fused{fused}_v{v}"] = {
                "ms": round(med, 4), "tflops": round(flops / med / 1e9, 1), "min_ms": round(min(times[v]), 4),
                "algorithmic_bytes_per_launch": alg}
            print(f"ks={ks} B={B} {ci}->{co} {H}x{W} fused={fused} variant={v}: {med:8.3f} ms  {flops / med / 1e9:7.1f} TFLOP/s", flush=True)
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
if os.environ.get("CALIB") == "1":
    from sgmse_amd import ops
    x = torch.randn(8, 128, 256, 512, device="cuda")
    w = torch.ones(128, device="cuda")
    for _ in range(3):
        ops.group_norm(x, w, w)
    torch.cuda.synchronize()
    res["_calibration"] = {"gn_chan_stats_kernel_read_bytes_per_launch": x.numel() * 4}
    # the HBM-bound kernel classes of the network under the same counters: FIR resamplers with the fused producer and both
    # outputs (as the residual blocks call them), algorithmic bytes = input once + outputs once
    sc, sh = torch.randn(8, 128, device="cuda"), torch.randn(8, 128, device="cuda")
    for up in (False, True):
        xin = x if not up else x[:, :, :128, :256].contiguous()
        for _ in range(3):
            ops.fir_resample(xin, up, in_scale=sc, in_shift=sh, in_act=True, return_raw=True)
        torch.cuda.synchronize()
        n_out = xin.numel() * (4 if up else 1) // (1 if up else 4)
        res["_calibration"]["fir_%s2_tiled_kernel_algorithmic_bytes_per_launch" % ("up" if up else "down")] = 4 * (xin.numel() + 2 * n_out)
json.dump(res, open(os.path.join(out_dir, os.environ.get("OUT", "conv_microbench.json")), "w"), indent=1)
