#!/usr/bin/env python
"""What does the vendor's own f16 GEMM reach on THIS box, on random data, and at what power and clock?

Context for the dominant kernel's roofline fraction: conv3x3_split_kernel issues 3 f16 MFMAs per
algorithmic multiply and is energy-bound at the 1400 W socket cap.  torch.matmul on fp16 tensors goes to hipBLASLt / rocBLAS:
a plain GEMM with no producer, no epilogue traffic to speak of and library-tuned tiles -- the practical ceiling of the f16
matrix pipe under the same cap.  Shapes: the implicit-GEMM shape of the dominant layer (M = 8 x 256 x 512 pixels, N = 128
output channels, K = 128 x 9, and K x 3 for the three partial products) and two square ones.  Each shape runs for ~3 s
while rocm-smi is sampled (measurement tool only; nothing here is on the product path)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import PowerSampler  # noqa: E402


def run(M, N, K, dtype, seconds=3.0, zeros=False):
    dev = "cuda"
    a = (torch.zeros if zeros else torch.randn)(M, K, device=dev, dtype=dtype)
    b = (torch.zeros if zeros else torch.randn)(K, N, device=dev, dtype=dtype)
    for _ in range(3):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.matmul(a, b); e1.record(); torch.cuda.synchronize()
    iters = max(10, int(seconds * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    with PowerSampler(period=0.25) as ps:
        t0 = time.time()
        e0.record()
        for _ in range(iters):
            torch.matmul(a, b)
        e1.record()
        torch.cuda.synchronize()
        wall = time.time() - t0
    ms = e0.elapsed_time(e1) / iters
    pw = ps.summary() or {}
    return {"M": M, "N": N, "K": K, "dtype": str(dtype).replace("torch.", ""), "data": "zeros" if zeros else "randn", "ms": round(ms, 4),
            "tflops": round(2.0 * M * N * K / ms / 1e9, 1), "frac_of_2500": round(2.0 * M * N * K / ms / 1e9 / 2500, 3),
            "power_w": round(pw.get("socket_power_w_mean") or 0, 0), "sclk_mhz": round(pw.get("sclk_mhz_mean") or 0, 0), "wall_s": round(wall, 2)}


def main():
    rows = []
    for (M, N, K) in [(8 * 256 * 512, 128, 1152), (8 * 256 * 512, 128, 3456), (8192, 8192, 8192), (16384, 16384, 4096)]:
        for dtype in (torch.float16, torch.bfloat16):
            rows.append(run(M, N, K, dtype))
            print(json.dumps(rows[-1]), flush=True)
    rows.append(run(8192, 8192, 8192, torch.float16, zeros=True))
    print(json.dumps(rows[-1]), flush=True)
    json.dump(rows, open(os.environ.get("OUT", "gpurun_out/gemm_reference.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
