"""Per-launch view of two kernel traces of tools/one_forward.py (tools/ab_trace.sh): the last forward of each, launch by launch."""
import csv, glob, re, sys
def load(d, k=-1):
    f = glob.glob(d + "/*/*_kernel_trace.csv")[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if "dec23_synth" in r["Kernel_Name"]]
    return [rows[ends[j - 1] + 1: ends[j] + 1] for j in range(len(ends) - 3, len(ends))]
A = load(sys.argv[1]); B = load(sys.argv[2])
n = len(A[0])
for i in range(n):
    ra = [f[i] for f in A]; rb = [f[i] for f in B]
    name = re.sub(r"\(.*", "", ra[0]["Kernel_Name"])[5:50]
    da = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ra)[1]
    db = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rb)[1]
    print("%-46s g%-8s q%-3s %7.1f %7.1f %+6.1f" % (name, ra[0]["Grid_Size_X"], ra[0]["Queue_Id"], da, db, db - da))
