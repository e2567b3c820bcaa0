#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one directory per counter pass) per kernel: mean counter values,
MFMA-pipe utilisation and effective clock.  Usage: summarize_pmc.py gpurun_out/pmc/p1 gpurun_out/pmc/p2 > out.json"""
import collections
import csv
import glob
import json
import sys

N_SIMD = 256 * 4
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, c in agg.items():
    if "conv" not in k[0] and "gn_chan" not in k[0] and "fir_" not in k[0]:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = {"dispatches": len(dur[k]), "mean_duration_us_under_pmc": sum(dur[k]) / len(dur[k]) / 1e3, "counters_mean": m}
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        act = m["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
        e["effective_clock_ghz"] = act / (e["mean_duration_us_under_pmc"] * 1e3)
        e["mfma_pipe_utilisation"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * N_SIMD)
    if "SQ_WAVE_CYCLES" in m:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in m:
                e[n.lower() + "_frac_of_wave_cycles"] = m[n] / m["SQ_WAVE_CYCLES"]
    out[f"{k[0]} grid={k[1]}"] = e
json.dump(out, sys.stdout, indent=1)
