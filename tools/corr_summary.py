#!/usr/bin/env python3
"""Per-shape summary of tools/one_corr.py under rocprofv3 (--kernel-trace and the two --pmc passes of prof_round.sh):
launch time, algorithmic bytes (2 C planes in, 81 planes out, N = 2), achieved GB/s against 8 TB/s.
usage: corr_summary.py <trace_dir> [fetch_dir write_dir]"""
import csv, glob, sys, json, collections
shapes = [(2, 196, 34, 60), (2, 128, 68, 120), (2, 96, 136, 240), (2, 64, 272, 480), (2, 32, 544, 960)]
def per_shape(d, key=None):
    f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    out = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "correlation_" not in r["Kernel_Name"]:
            continue
        g = r.get("Grid_Size") or r.get("Grid_Size_X") or r.get("Workgroup_Count") or "?"
        if key is None:
            out.setdefault(g, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        elif r["Counter_Name"] == key:
            out.setdefault(g, {}).setdefault(r["Dispatch_Id"], 0.0)
            out[g][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return out
t = per_shape(sys.argv[1])
fe = per_shape(sys.argv[2], "FETCH_SIZE") if len(sys.argv) > 3 else {}
wr = per_shape(sys.argv[3], "WRITE_SIZE") if len(sys.argv) > 3 else {}
rows = []
for (g, v), (n, c, h, w) in zip(sorted(t.items(), key=lambda kv: float(kv[0]) if kv[0] != "?" else 0), shapes):
    us = sum(v) / len(v)
    alg = (2 * c + 81) * n * h * w * 4
    row = {"shape_NCHW": [n, c, h, w], "launch_us": round(us, 1), "algorithmic_MB": round(alg / 1e6, 1), "achieved_GBps": round(alg / us / 1e3, 1),
           "frac_of_8TBps": round(alg / us / 1e3 / 8000, 3)}
    if g in fe: row["fetch_x2_MB"] = round(2 * 1024 * sum(fe[g].values()) / len(fe[g]) / 1e6, 1)
    if g in wr: row["write_MB"] = round(1024 * sum(wr[g].values()) / len(wr[g]) / 1e6, 1)
    rows.append(row)
print(json.dumps({"kernel": "correlation_dma_kernel<8> (PWC-Net cost volume, radius 4, 81 channels; LDS-DMA loader waves)", "per_shape": rows}, indent=1))
