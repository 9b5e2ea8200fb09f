"""Fold the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of tools/one_conv_spk.py
into profiles/r01_conv96_spk_traffic.json.  usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, sys
def per_launch(d, counter):
    f = sorted(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime)[-1]          # the newest: older calls' files are merged into the same directory
    vals = {}
    for r in csv.DictReader(open(f)):
        if "conv3x3_spk_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.setdefault(r["Dispatch_Id"], 0.0)
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = sorted(vals.values())
    return v[len(v) // 2], len(v)
fetch, n1 = per_launch(sys.argv[1], "FETCH_SIZE")
write, n2 = per_launch(sys.argv[2], "WRITE_SIZE")
h, w = 288, 480
alg = 96 * h * w * 4 * 2          # packed in (4 B/element) + packed out
out = {"kernel": "conv3x3_spk_kernel<3,3,false> 96->96 3x3 @288x480 (split-packed in and out)",
       "FETCH_SIZE_KiB_raw_median": fetch, "WRITE_SIZE_KiB_median": write, "launches": [n1, n2],
       "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B for 16-B/lane streams, which every load of this kernel is: "
                     "MI355X_MICROARCH.md, HBM); Infinity-Cache hits are included in the counter; separate --pmc passes",
       "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
       "algorithmic_bytes_per_launch": alg,
       "note": "the counter sits on the L2's fabric side: the 331 KB of split weights are re-fetched by every XCD's L2 and each input tile by "
               "both output-channel groups; reads beyond the algorithmic 53 MB are L2 misses served by the Infinity Cache, not HBM re-reads"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
