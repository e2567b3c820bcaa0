#!/usr/bin/env python3
"""Counters of the bench command's OWN launches (tools/gpu_visit.sh STEPS=pmcbench): rocprofv3 --pmc passes over
`bench.py --steps 1 --warmup 1 --no-others --no-cpu-baseline --no-profile` (one pass per counter group; FETCH_SIZE and
WRITE_SIZE in separate passes as MI355X_MICROARCH.md prescribes: both in units of 1024 bytes), summarised per kernel
instantiation over every launch of the run.

    summarize_pmc_bench.py <pass dir> [<pass dir> ...] [--dump prof_dump_b32.txt] [--meta key=value ...] > profiles/rNN_pmc_bench_b32.json

Round 6: (i) CALIBRATED traffic -- the passes run the bench command with SGMSE_PMC_CALIB=1, which puts four launches of known size in
front of the run (calib_stream_kernel<VEC, WRITE>: a coalesced 1 GiB read / write stream at 8 and at 16 bytes per lane); factor = known
bytes / (counter x 1024) per width; the convolution kernels get the 8-byte factors (input pairs, output pairs; their 16-byte
weight fragments are < 1 % of a launch's bytes and the 16-byte factor is recorded beside them), raw values kept beside the calibrated ones;
(ii) `_meta` (command, workload, batch, commit, hash of the kernel sources) so that bench.py applies the file only to the command it was
taken from; (iii) every kernel the include-regex let through is summarised, attn_core_kernel among them (MFMA utilisation of attention).

Per kernel: launches, mean duration under the counters, mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x
GRBM_GUI_ACTIVE / 8 XCDs), effective clock, instruction mix, LDS conflict share, HBM bytes per launch.  --dump: a
SGMSE_PROFILE_DUMP=1 listing of one evaluation at the same batch; the algorithmic bytes of the Winograd launches (input once +
output once + residual / shortcut input once + weights) are averaged per instantiation from its lines."""
import collections
import csv
import glob
import json
import re
import sys

N_SIMD = 256 * 4
dirs, dump, meta = [], None, {}
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--dump":
        dump = args.pop(0)
    elif a == "--meta":
        while args and "=" in args[0] and not args[0].startswith("--"):
            k, v = args.pop(0).split("=", 1)
            meta[k] = int(v) if (k == "batch" and v.isdigit()) else v
    else:
        dirs.append(a)
CALIB_BYTES = float(1 << 30)          # bench.py: CALIB_BYTES

vals = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> per-dispatch values
dur = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

# algorithmic bytes of the Winograd launches from the per-launch listing: instantiation <8|4, act, sc>
alg, alg_rd = collections.defaultdict(list), collections.defaultdict(list)
if dump:
    pat = re.compile(r"conv3x3-wino (\d+)->(\d+) @(\d+)x(\d+)x(\d+)(.*?) ([\d.]+) ms")
    for line in open(dump):
        m = pat.search(line)
        if not m:
            continue
        ci, co, B, H, W = (int(m.group(i)) for i in range(1, 6))
        flags = m.group(6)
        px = B * H * W
        sc = "+shortcut" in flags
        nblk8 = B * ((H + 7) // 8) * ((W + 31) // 32) * ((co + 127) // 128)
        rows = 8 if nblk8 >= 512 else 4
        bytes_ = 4.0 * px * (ci + co) + 4.0 * co * ci * 9
        if "+res" in flags:
            bytes_ += 4.0 * px * co
        if sc:
            ms = re.search(r"\+shortcut\((\d+)\)", flags)
            bytes_ += 4.0 * px * (int(ms.group(1)) if ms else co)     # the folded 1x1 shortcut reads the block's raw input
        name = f"sgmse::conv3x3_wino_kernel<{rows}, {1 if '+gn' in flags else 0}, {1 if sc else 0}, 0, 0>"
        alg[name].append(bytes_)
        alg_rd[name].append(bytes_ - 4.0 * px * co)

# calibration rows of the same passes: calib_stream_kernel<VEC floats per lane, WRITE>
calib = {"known_bytes_per_launch": CALIB_BYTES, "source": "calib_stream_kernel launches in the same rocprofv3 pass (bench.py SGMSE_PMC_CALIB=1)"}
for k in vals:
    m = re.search(r"calib_stream_kernel<(\d+), (\d+)>", k)
    if not m:
        continue
    width, wr = 4 * int(m.group(1)), int(m.group(2))
    cname = "WRITE_SIZE" if wr else "FETCH_SIZE"
    if vals[k].get(cname):
        rep = sum(vals[k][cname]) / len(vals[k][cname]) * 1024.0
        calib[f"{'write' if wr else 'fetch'}_{width}B_per_lane"] = {"reported_bytes": rep, "factor": CALIB_BYTES / rep if rep > 0 else None}
f8 = (calib.get("fetch_8B_per_lane") or {}).get("factor")
w8 = (calib.get("write_8B_per_lane") or {}).get("factor")
calib["applied_to_conv_kernels"] = {"fetch": f8 or 2.0, "write": w8 or 1.0,
                                    "note": ("fetch / write factors of the 8-byte-per-lane streams (the staging loads and the output stores of the "
                                             "Winograd kernel are 8-byte column pairs)" if f8 else
                                             "no calibration rows in the passes: the guide's factor 2 for coalesced reads, 1 for writes")}

out = {"_meta": meta, "_calibration": calib}
for k in sorted(vals, key=lambda n: -sum(dur[n])):
    c = {n: sum(v) / len(v) for n, v in vals[k].items()}
    n_launch = max(len(v) for v in vals[k].values())
    e = {"launches": n_launch, "mean_duration_us_under_pmc": sum(dur[k]) / len(dur[k]) / 1e3, "total_ms_under_pmc": sum(dur[k]) / 1e6 / max(1, len(dirs))}
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        act = c["GRBM_GUI_ACTIVE"] / 8.0
        e["effective_clock_ghz"] = act / (e["mean_duration_us_under_pmc"] * 1e3)
        e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * N_SIMD)
    if "SQ_INSTS_VALU" in c and c.get("SQ_INSTS_MFMA"):
        e["valu_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        fr, wr = c.get("FETCH_SIZE", 0.0) * 1024.0, c.get("WRITE_SIZE", 0.0) * 1024.0
        ff, wf = ((calib["applied_to_conv_kernels"]["fetch"], calib["applied_to_conv_kernels"]["write"]) if "calib_stream" not in k else (1.0, 1.0))
        e["fetch_bytes_per_launch_raw"], e["write_bytes_per_launch_raw"] = fr, wr
        e["fetch_bytes_per_launch"], e["write_bytes_per_launch"] = fr * ff, wr * wf
        e["hbm_bytes_per_launch"] = fr * ff + wr * wf
    if k in alg:
        e["algorithmic_bytes_per_launch"] = sum(alg[k]) / len(alg[k])
        e["algorithmic_read_bytes_per_launch"] = sum(alg_rd[k]) / len(alg_rd[k])
        e["launches_per_eval_in_the_listing"] = len(alg[k])
        if e.get("hbm_bytes_per_launch"):
            e["traffic_vs_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
            e["fetch_vs_algorithmic_reads"] = e["fetch_bytes_per_launch"] / e["algorithmic_read_bytes_per_launch"]
    e["counters_mean"] = c
    out[k] = e
json.dump(out, sys.stdout, indent=1)
