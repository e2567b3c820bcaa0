#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per-kernel totals AND the idle time between consecutive kernels (launch gaps), which is what
matters at batch 1.  Usage: summarize_kernel_trace.py <dir or *_kernel_trace.csv> [skip_first_fraction]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * skip):]          # the later part of the run: steady-state graph replays


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("sgmse::", "")
    return n[:70]


busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 200_000]
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} kernels over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms ({busy / span:.1%}), gaps < 0.2 ms: {sum(small) / 1e6:.2f} ms "
      f"(mean {sum(small) / max(len(small), 1) / 1e3:.2f} us), larger gaps: {sum(g for g in gaps if g >= 200_000) / 1e6:.2f} ms")
agg = defaultdict(lambda: [0, 0, 0])
for i, (s, e, n) in enumerate(rows):
    a = agg[short(n)]
    a[0] += 1
    a[1] += e - s
    if i < len(gaps) and gaps[i] < 200_000:
        a[2] += gaps[i]
print(f"{'kernel':72s} {'calls':>7s} {'total ms':>9s} {'avg us':>8s} {'gap after, avg us':>18s}")
for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} {c:7d} {d / 1e6:9.3f} {d / c / 1e3:8.2f} {g / c / 1e3:18.2f}")
