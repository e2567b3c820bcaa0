#!/usr/bin/env python3
"""Do the network's bits depend on what ELSE runs on the GPU?  (round 6: the side-stream experiment produced run-to-run differences
when two of the engine's launches overlapped; this checks the product path -- one stream per context -- under outside load.)
Process A: full-width forward, B = 1, T = 64 and T = 512, N repetitions, each compared with A's own first result.
Process B (optional, `--load`): another context hammering the device with forwards at batch 4 the whole time.
    python tools/probes/concurrency_bits_probe.py            # spawns the load process itself, prints per-shape verdicts
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch


def model():
    from parity import NET_CASES, make_backbone
    from sgmse_amd import _lib
    _lib.load_library()
    return make_backbone(NET_CASES[os.environ.get("PROBE_NET", "fwd_nf128")], "cuda")[0]


if "--load" in sys.argv:
    net = model()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(4, 2, 256, 128, dtype=torch.complex64, generator=g) * 0.3).cuda()
    t = torch.full((4,), 0.5).cuda()
    net(x, t); torch.cuda.synchronize()
    print("LOAD-RUNNING", flush=True)              # (tests wait for this line: the device is loaded from here on)
    t0 = time.time()
    stop = os.environ.get("LOAD_STOP_FILE")        # (ended cleanly by whoever started it: the file appears)
    while time.time() - t0 < float(os.environ.get("LOAD_SECONDS", "60")) and not (stop and os.path.exists(stop)):
        net(x, t)
    torch.cuda.synchronize()
    sys.exit(0)

net = model()
g = torch.Generator().manual_seed(3)
PB = int(os.environ.get("PROBE_B", "1"))
cases = {T: ((torch.randn(PB, 2, 256, T, dtype=torch.complex64, generator=g) * 0.3).cuda(), torch.full((PB,), 0.4).cuda()) for T in [int(v) for v in os.environ.get("PROBE_T", "64,512").split(",")]}
ref = {T: net(x, t).cpu() for T, (x, t) in cases.items()}
solo = {T: all(torch.equal(net(x, t).cpu(), ref[T]) for _ in range(5)) for T, (x, t) in cases.items()}
print("alone on the device: repeated forwards identical:", solo, flush=True)
load = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--load"], env=dict({k: v for k, v in os.environ.items() if not k.startswith("SGMSE_") and k != "PROBE_NET"}, LOAD_SECONDS=os.environ.get("LOAD_SECONDS", "45")))
time.sleep(12)            # (the load process needs its library load and first forward)
n = int(os.environ.get("REPS", "40"))
for T, (x, t) in cases.items():
    bad = 0
    worst = 0.0
    for _ in range(n):
        o = net(x, t).cpu()
        if not torch.equal(o, ref[T]):
            bad += 1
            worst = max(worst, float((o - ref[T]).abs().max()))
    print(f"under outside load, T = {T}: {n - bad} of {n} forwards identical to the solo result" + (f"; worst |diff| {worst:.2e}" if bad else ""), flush=True)
load.wait()
