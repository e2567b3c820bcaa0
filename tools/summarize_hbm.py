#!/usr/bin/env python3
"""HBM traffic per launch of every kernel from two rocprofv3 PMC passes over the SAME command (FETCH_SIZE in one pass,
WRITE_SIZE in the other; TCC slots do not fit both).  Units and corrections follow MI355X_MICROARCH.md (HBM):
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B, and on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide (16 B/lane) coalesced read stream -> multiplied by 2 for the float4-staged kernels (marked below).
Usage: summarize_hbm.py <fetch_dir> <write_dir> > profiles/rNN_hbm_traffic.json"""
import collections
import csv
import glob
import json
import sys

WIDE_READERS = ("conv_mfma_kernel", "gn_chan_stats_kernel")   # kernels whose read stream is 16 B/lane


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write), key=lambda n: -sum(fetch.get(n, [0]))):
    f, w = fetch.get(k, []), write.get(k, [])
    corr = 2.0 if any(s in k for s in WIDE_READERS) else 1.0
    rd = sum(f) / max(len(f), 1) * 1024.0 * corr
    wr = sum(w) / max(len(w), 1) * 1024.0
    out[k] = {"launches": len(f), "fetch_bytes_per_launch": rd, "fetch_correction": corr, "write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": rd + wr}
json.dump(out, sys.stdout, indent=1)
