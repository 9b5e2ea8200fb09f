#!/usr/bin/env python3
"""Fold the three rocprofv3 passes of tools/prof_forward_pmc.sh into one per-kernel table: launches and GPU time per
forward (kernel trace), HBM-side read / written bytes per forward (FETCH_SIZE / WRITE_SIZE, KiB units; separate --pmc
passes).  FETCH_SIZE is reported raw AND corrected: on gfx950 it tallies 128-byte requests at 64 bytes for wide (16 B per
lane) streaming reads (MI355X_MICROARCH.md, HBM); the plane_bw_bench calibration passes (known byte counts at 4 B and
16 B per lane) say which factor applies to which access width.  usage: pmc_forward_summary.py <dir> <n_forwards>"""
import csv, glob, json, re, sys
from collections import OrderedDict

d, nf = sys.argv[1], float(sys.argv[2])


def short(name):
    n = re.sub(r"^void ", "", name)
    n = n.split("(")[0]
    return n[-70:]


def counters(sub, counter):
    fs = glob.glob("%s/%s/*/*counter_collection.csv" % (d, sub))
    out = OrderedDict()
    if not fs:
        return out
    per = {}
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != counter:
            continue
        k = (short(r["Kernel_Name"]), r["Dispatch_Id"])
        per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
    for (n, _), v in per.items():
        e = out.setdefault(n, [0, 0.0])
        e[0] += 1
        e[1] += v
    return out


dur = OrderedDict()
fs = glob.glob(d + "/trace/*/*kernel_trace.csv")
for r in csv.DictReader(open(fs[0])):
    e = dur.setdefault(short(r["Kernel_Name"]), [0, 0.0])
    e[0] += 1
    e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")

cal = {}
cf, cw = counters("cal_fetch", "FETCH_SIZE"), counters("cal_write", "WRITE_SIZE")
HW = 2304 * 3840
cfgs = [(1, 1), (6, 0), (6, 3), (26, 0), (26, 4), (6, 18), (0, 18), (18, 3)]
for name, mode in (("k<1, 4>", "4B_64x4"), ("k<1, 1>", "4B_256x1"), ("k<4, 1>", "16B_1024x1")):
    if name in cf:
        rd = sum(p for p, q in cfgs) * HW * 4 * 3          # 3 repetitions per configuration
        wr = sum(q for p, q in cfgs) * HW * 4 * 3
        cal[mode] = {"fetch_reported_over_true": round(cf[name][1] * 1024 / rd, 4),
                     "write_reported_over_true": round(cw[name][1] * 1024 / wr, 4) if name in cw else None}

rows = []
for n, (c, us) in dur.items():
    if us / nf < 1.0 and n not in fetch:
        continue
    f = fetch.get(n, [0, 0.0])[1] * 1024 / nf
    w = write.get(n, [0, 0.0])[1] * 1024 / nf
    rows.append({"kernel": n, "launches_per_forward": round(c / nf, 2), "us_per_forward": round(us / nf, 1),
                 "avg_us": round(us / c, 2), "fetch_raw_MB": round(f / 1e6, 1), "fetch_x2_MB": round(2 * f / 1e6, 1),
                 "write_MB": round(w / 1e6, 1)})
rows.sort(key=lambda r: -r["us_per_forward"])
# One-time weight prepacks (absmax / prepack kernels run on the first forward only) and the harness's torch operators (bicubic
# pyramid, reflection pad, fills, copies: what one_forward.py does AROUND the forward) are not part of a warm forward: listed apart,
# outside the totals.
ONE_TIME = re.compile(r"prepack|absmax|at::native|__amd_rocclr|FillFunctor|direct_copy|elementwise_kernel|pcap_init_kernel_once")
fwd = [r for r in rows if not ONE_TIME.search(r["kernel"])]
other = [r for r in rows if ONE_TIME.search(r["kernel"])]
tot = sum(r["us_per_forward"] for r in fwd)
launches = sum(r["launches_per_forward"] for r in fwd)
moved = sum(r["fetch_x2_MB"] + r["write_MB"] for r in fwd)
json.dump({"n_forwards": nf, "calibration": cal, "gpu_us_per_forward": round(tot, 1), "launches_per_forward": round(launches, 1),
           "moved_MB_per_forward": round(moved, 1), "kernels": fwd, "one_time_or_harness": other},
          open(d + "/summary.json", "w"), indent=1)
print("calibration (reported / true bytes):", json.dumps(cal))
print("warm forward: %.1f us of GPU time, %.0f launches, %d kernel kinds; HBM-side bytes moved (FETCH x 2 + WRITE): %.1f MB"
      % (tot, launches, len(fwd), moved))
hdr = "%-72s %6s %9s %8s %10s %10s %9s" % ("kernel", "n/fwd", "us/fwd", "avg us", "fetch MB", "fetchx2 MB", "write MB")
print(hdr)
for r in fwd:
    print("%-72s %6.1f %9.1f %8.2f %10.1f %10.1f %9.1f" % (r["kernel"], r["launches_per_forward"], r["us_per_forward"], r["avg_us"],
                                                           r["fetch_raw_MB"], r["fetch_x2_MB"], r["write_MB"]))
if other:
    print("-- not part of a warm forward (one-time weight prepacks, the harness's torch pyramid / copies), per profiled forward:")
    for r in other:
        print("%-72s %6.1f %9.1f %8.2f %10.1f %10.1f %9.1f" % (r["kernel"], r["launches_per_forward"], r["us_per_forward"], r["avg_us"],
                                                               r["fetch_raw_MB"], r["fetch_x2_MB"], r["write_MB"]))
