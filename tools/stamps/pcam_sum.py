import csv,glob,sys
for d in sys.argv[1:]:
    f=glob.glob(d+"/*/*kernel_stats.csv")[0]
    for r in csv.DictReader(open(f)):
        if "pcam_kernel" in r["Name"]: print(d.split("/")[-1], r["Name"][:28], r["Calls"], "avg %.1f us min %.1f"%(float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
