#!/usr/bin/env python3
"""Phase time stamps of conv3x3_split_kernel<..., ABL = 64> (16 x u64 per workgroup: hw_id | xcc << 32, t_start, t_loop,
t_epilogue, t_end; shader clock): per-phase durations and, per CU, how the resident workgroups' phases interleave."""
import collections
import sys

import numpy as np

WINO = len(sys.argv) > 2 and sys.argv[2] == "wino"
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 32 if WINO else 16)
a = a[a[:, 1] > 0]
hw = (a[:, 0] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (a[:, 0] >> np.uint64(32)).astype(np.int64) & 0xf
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
t = a[:, 1:5].astype(np.int64)
t0 = t[:, 0].min()
t = t - t0
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
print(f"{len(a)} workgroups; kernel span {t[:, 3].max()} cycles")
for name, v in (("prologue", pro), ("K loop", loop), ("epilogue", epi), ("total", t[:, 3] - t[:, 0])):
    print(f"{name:9s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}")
ts = a[:, 5:15].astype(np.int64)
if len(sys.argv) > 2 and sys.argv[2] == "wino":      # conv3x3_wino_kernel<..., TRACE>: [5] = LDS exchange, [6] = time at the stage barriers
    print(f"output-transform exchange mean {ts[:, 0].mean():9.0f}; stage barriers (thread 0) mean {ts[:, 1].mean():9.0f} cycles")
    print("epilogue of wave 0, cycles from the end of the K loop: residual loads issued %.0f, partial sums written %.0f, barrier passed %.0f, "
          "combined %.0f, outputs stored %.0f, end %.0f" % (ts[:, 2].mean(), ts[:, 3].mean(), ts[:, 4].mean(), ts[:, 0].mean(), ts[:, 5].mean(), epi.mean()))
    for name, off in (("A-wave (thread 0)", 16), ("B-wave (thread 256)", 23)):
        tt = a[:, off:off + 7].astype(np.int64)
        print(f"K loop of the {name}: cycles per tap position summed over the stages " + " ".join(f"{tt[:, i].mean():.0f}" for i in range(6)) +
              f"; at the stage barriers {tt[:, 6].mean():.0f}")
elif ts.sum() > 0:
    tot = ts.sum(axis=1).mean()
    print("K loop of wave 0, mean time per tap position summed over the stages (share of the loop):")
    for i in range(9):
        print(f"   tap {i}: {ts[:, i].mean():9.0f} ({ts[:, i].mean() / tot:5.1%})")
    print(f"   stage barrier: {ts[:, 9].mean():9.0f} ({ts[:, 9].mean() / tot:5.1%})")
key = xcc * 4096 + se * 256 + sh * 16 + cu
groups = collections.defaultdict(list)
for i, k in enumerate(key):
    groups[int(k)].append(i)
print(f"{len(groups)} distinct (xcc, se, sh, cu) placements; workgroups per placement: min {min(map(len, groups.values()))} max {max(map(len, groups.values()))}")
# overlap statistics: fraction of a workgroup's epilogue time during which another workgroup of the same CU is in its K loop
ov_loop, ov_epi = [], []
for k, idx in groups.items():
    idx = sorted(idx, key=lambda i: t[i, 0])
    for i in idx:
        e0, e1 = t[i, 2], t[i, 3]
        if e1 <= e0:
            continue
        in_loop = in_epi = 0
        for j in idx:
            if j == i:
                continue
            in_loop += max(0, min(e1, t[j, 2]) - max(e0, t[j, 1]))
            in_epi += max(0, min(e1, t[j, 3]) - max(e0, t[j, 2]))
        ov_loop.append(in_loop / (e1 - e0)); ov_epi.append(in_epi / (e1 - e0))
print(f"during a workgroup's epilogue, a co-resident workgroup is in its K loop {np.mean(ov_loop):.2f} of the time, in its own epilogue {np.mean(ov_epi):.2f}")
k0 = sorted(groups)[0]
print(f"timeline of placement {k0:#x} (start, loop, epilogue, end; cycles from the first start):")
for i in sorted(groups[k0], key=lambda i: t[i, 0])[:12]:
    print("   ", t[i].tolist(), "simd/wave", (hw[i] >> 4) & 3, hw[i] & 0xf)
