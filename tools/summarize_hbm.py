#!/usr/bin/env python3
"""HBM traffic per launch of every kernel from two rocprofv3 PMC passes over the SAME command (FETCH_SIZE in one pass,
WRITE_SIZE in the other; the TCC block cannot hold both).  Units and corrections follow MI355X_MICROARCH.md (HBM):
both counters are in units of 1024 bytes; on gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced read streams
(by exactly 2x in the guide's measurement), so the read side is CALIBRATED in the same run on a kernel with a known
float4 read stream (gn_chan_stats_kernel, byte count recorded by tools/conv_microbench.py CALIB=1) and that factor is
applied to the float4-staged convolution kernels.  WRITE_SIZE matched known byte counts exactly (fill_random_kernel).
Usage: summarize_hbm.py <fetch_dir> <write_dir> [microbench.json] > profiles/rNN_hbm_traffic.json"""
import collections
import csv
import glob
import json
import sys


DUR = collections.defaultdict(list)      # kernel -> launch durations (ns) under the counter passes


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0]
                acc[k].append(float(r["Counter_Value"]))
                DUR[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
micro = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
calib = micro.get("_calibration", {})
corr, note = 2.0, "guide value (FETCH_SIZE = half of a 16 B/lane coalesced stream)"
known = calib.get("gn_chan_stats_kernel_read_bytes_per_launch")
gn = [k for k in fetch if "gn_chan_stats_kernel" in k]
if known and gn:
    big = [v for v in fetch[gn[0]] if v * 1024 > 0.2 * known]      # launches over the calibration tensor
    if big:
        corr = known / (sum(big) / len(big) * 1024.0)
        note = f"calibrated in this run on gn_chan_stats_kernel ({known} known bytes per launch)"
out = {"_read_correction": {"factor": corr, "source": note}}
for k in sorted(set(fetch) | set(write), key=lambda n: -sum(fetch.get(n, [0]))):
    f, w = fetch.get(k, []), write.get(k, [])
    c = corr if any(n in k for n in ("conv_mfma_kernel", "gn_chan_stats_kernel", "fir_down2_tiled", "fir_up2_tiled", "gn_finalize")) else 1.0
    rd = sum(f) / max(len(f), 1) * 1024.0 * c
    wr = sum(w) / max(len(w), 1) * 1024.0
    out[k] = {"launches": len(f), "fetch_bytes_per_launch": rd, "fetch_correction": c, "write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": rd + wr}
    if DUR.get(k):
        us = sorted(DUR[k])[len(DUR[k]) // 2] / 1e3
        out[k]["median_duration_us_under_pmc"] = us
        out[k]["hbm_gb_per_s"] = (rd + wr) / (us * 1e-6) / 1e9 if us > 0 else None
for name, val in calib.items():
    if name.endswith("_algorithmic_bytes_per_launch"):
        kern = name[:-len("_algorithmic_bytes_per_launch")]
        for k in out:
            if kern in k and isinstance(out[k], dict):
                out[k]["algorithmic_bytes_per_launch"] = val
                if out[k].get("median_duration_us_under_pmc"):
                    out[k]["algorithmic_gb_per_s"] = val / (out[k]["median_duration_us_under_pmc"] * 1e-6) / 1e9
if False:
    pass
alg = [v["algorithmic_bytes_per_launch"] for k, v in micro.items() if not k.startswith("_")]
if alg:
    out["_algorithmic_bytes_per_launch_of_the_benchmarked_conv"] = sum(alg) / len(alg)
json.dump(out, sys.stdout, indent=1)
