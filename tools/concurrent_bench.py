#!/usr/bin/env python3
"""Two (or more) bench.py processes on the ONE GPU at the same time, against one process alone: does overlapping one utterance
stream's light phases (FIR, epilogues, coarse levels) with another's matrix-heavy ones raise the device's total rate?
    python tools/concurrent_bench.py BATCH NPROC [STEPS]      -> one line per arm"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch, nproc = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(batch), "--steps", str(steps), "--warmup", "1", "--no-others",
       "--no-cpu-baseline", "--no-profile"]


def line(p):
    out, err = p.communicate(timeout=1200)
    rows = [l for l in out.splitlines() if l.startswith("{")]
    if p.returncode != 0 or len(rows) != 1:
        raise SystemExit(out[-2000:] + err[-2000:])
    return json.loads(rows[0])


t0 = time.perf_counter()
procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT) for _ in range(nproc)]
rows = [line(p) for p in procs]
print(json.dumps({"batch_per_process": batch, "processes": nproc, "steps": steps, "utt_per_s_each": [round(r["value"], 4) for r in rows],
                  "utt_per_s_together": round(sum(r["value"] for r in rows), 4), "ms_per_step_each": [round(r["ms_per_step"], 1) for r in rows],
                  "wall_s": round(time.perf_counter() - t0, 1)}))
