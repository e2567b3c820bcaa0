#!/usr/bin/env python3
"""Calibration of bench.py's cpu_baseline (kind "port") against the reference module it restates.  BUILD CONTAINER ONLY: imports the
reference's NCSNpp from /root/reference (absent on the GPU box) and the oracle port, runs both on the same [1,4,256,512] evaluation,
interleaved, and prints seconds per evaluation and their ratio (bench.py: PORT_VS_REFERENCE).

    python tools/port_vs_reference.py <threads> <repeats>
"""
import os
import statistics as st
import sys
import time

os.environ["HIP_VISIBLE_DEVICES"] = ""
os.environ["CUDA_VISIBLE_DEVICES"] = ""
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("SGMSE_REFERENCE", "/root/reference"))
import torch  # noqa: E402

from oracle import ncsnpp_oracle as NO, synth  # noqa: E402
from oracle.make_golden import ref_model  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.set_num_threads(threads)
cfg = NO.NetCfg.for_variant("ncsnpp")
P = synth.synth_params(cfg, seed=0)
m = ref_model(cfg, P)
g = torch.Generator().manual_seed(11)
x = torch.randn(1, 2, 256, 512, dtype=torch.complex64, generator=g) * 0.3
t = torch.tensor([0.5])
with torch.no_grad():
    m(x, t)
    NO.ncsnpp_forward(P, cfg, x, t)
    tr, tp = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); m(x, t); tr.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); NO.ncsnpp_forward(P, cfg, x, t); tp.append(time.perf_counter() - t0)
print(f"threads {threads}: reference {st.median(tr):.3f} s, port {st.median(tp):.3f} s per evaluation, ratio {st.median(tr) / st.median(tp):.3f}")
