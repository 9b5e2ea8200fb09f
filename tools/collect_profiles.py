#!/usr/bin/env python3
"""Condense one `tools/prof_round.sh <tag>` output directory (gpurun_out/<tag>) into the tracked files under profiles/:

  <prefix>_forward_pmc.{txt,json}          per-kernel forward table (duration, FETCH/WRITE, bytes vs algorithmic)
  <prefix>_forward_kernel_stats.csv        rocprofv3 --stats of the same forward
  <prefix>_conv96_ring_kernel_stats.csv    rocprofv3 --stats of the dominant conv alone
  <prefix>_conv96_ring_sq.txt              SQ / GRBM counter medians of the dominant conv
  <prefix>_conv96_spk_traffic.json         FETCH/WRITE medians, corrected HBM bytes per launch (bench.py reads this one)
  <prefix>_correlation.json                stand-alone PWC correlation: duration, traffic, GB/s
  <prefix>_bench_kernel_stats.csv          rocprofv3 --stats of the bench command itself
  <prefix>_bench_under_rocprof.json        the bench line of that profiled run

usage: collect_profiles.py <tag> [prefix]        (prefix defaults to r02)
"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    """The NEWEST match: gpurun merges every call's files into the same local directories (rocprofv3 names them by process id), so an older run's
    files lie beside the current one's."""
    f = sorted(glob.glob(pattern), key=os.path.getmtime)
    return f[-1] if f else None


def pmc_medians(d, substr):
    f = one(d + "/*/*counter_collection.csv")
    if not f:
        return {}
    vals = {}
    for r in csv.DictReader(open(f)):
        if substr in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            vals[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    out = {}
    for c, dd in vals.items():
        v = sorted(dd.values())
        out[c] = (v[len(v) // 2], len(v))
    return out


def stats_row(d, substr):
    f = one(d + "/*/*kernel_stats.csv")
    if not f:
        return None
    for r in csv.DictReader(open(f)):
        if substr in r["Name"]:
            return r
    return None


def copy(src, dst):
    if src and os.path.exists(src):
        shutil.copyfile(src, dst)
        print("wrote", os.path.relpath(dst, ROOT))
        return True
    print("MISSING", src)
    return False


def main():
    tag = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else "r02"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    p = lambda name: os.path.join(dst, prefix + "_" + name)

    copy(src + "/fwd_pmc/summary.txt", p("forward_pmc.txt"))
    copy(src + "/fwd_pmc/summary.json", p("forward_pmc.json"))
    copy(one(src + "/fwd_pmc/trace/*/*kernel_stats.csv"), p("forward_kernel_stats.csv"))
    copy(one(src + "/conv96_trace/*/*kernel_stats.csv"), p("conv96_ring_kernel_stats.csv"))
    copy(one(src + "/bench_trace/*/*kernel_stats.csv"), p("bench_kernel_stats.csv"))
    copy(src + "/bench_under_rocprof.json", p("bench_under_rocprof.json"))

    # dominant conv: SQ / GRBM medians and the traffic record
    kern = "conv3x3_ring_kernel"
    sq = pmc_medians(src + "/conv96_sq", kern)
    row = stats_row(src + "/conv96_trace", kern)
    if sq:
        with open(p("conv96_ring_sq.txt"), "w") as f:
            f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/one_conv_spk.py (REPS=20)\n")
            f.write("# median per launch of the 96->96 3x3 ring kernel at 288x480 (N = 6 planes of the level-0 UNet: one sample)\n")
            for c, (v, n) in sorted(sq.items()):
                f.write("%-32s %16.0f   (%d launches)\n" % (c, v, n))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "SQ_BUSY_CYCLES" in sq and sq["SQ_BUSY_CYCLES"][0] > 0:
                f.write("MFMA busy / SQ busy               %16.3f\n" % (sq["SQ_VALU_MFMA_BUSY_CYCLES"][0] / sq["SQ_BUSY_CYCLES"][0]))
            if row:
                f.write("average duration (kernel trace)  %16.1f us over %s calls\n" % (float(row["AverageNs"]) / 1e3, row["Calls"]))
        print("wrote", os.path.relpath(p("conv96_ring_sq.txt"), ROOT))
    fe = pmc_medians(src + "/conv96_fetch", kern).get("FETCH_SIZE")
    wr = pmc_medians(src + "/conv96_write", kern).get("WRITE_SIZE")
    if fe and wr:
        rec = {
            "kernel": "conv3x3_ring_kernel<3,3,false,8> 96->96 3x3 @288x480 (split-packed in and out)",
            "FETCH_SIZE_KiB_raw_median": fe[0], "WRITE_SIZE_KiB_median": wr[0], "launches": [fe[1], wr[1]],
            "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B; calibrated against a known-size copy in "
                          "tools/prof_forward_pmc.sh: factor 0.50 for 4-B and 16-B lanes); WRITE_SIZE exact; Infinity-Cache hits are "
                          "included in the counter; separate --pmc passes",
            "hbm_bytes_per_launch": int((2 * fe[0] + wr[0]) * 1024),
            "algorithmic_bytes_per_launch": 106168320,
            "average_duration_us": float(row["AverageNs"]) / 1e3 if row else None,
        }
        json.dump(rec, open(p("conv96_spk_traffic.json"), "w"), indent=1)
        print("wrote", os.path.relpath(p("conv96_spk_traffic.json"), ROOT), rec["hbm_bytes_per_launch"])

    # correlation alone: per-shape table (tools/corr_summary.py keys launches by grid size)
    if one(src + "/corr_trace/*/*kernel_trace.csv"):
        import subprocess
        args = [sys.executable, os.path.join(ROOT, "tools", "corr_summary.py"), src + "/corr_trace"]
        if one(src + "/corr_fetch/*/*counter_collection.csv") and one(src + "/corr_write/*/*counter_collection.csv"):
            args += [src + "/corr_fetch", src + "/corr_write"]
        r = subprocess.run(args, capture_output=True, text=True)
        if r.returncode == 0:
            open(p("correlation.json"), "w").write(r.stdout)
            print("wrote", os.path.relpath(p("correlation.json"), ROOT))
        else:
            print("corr_summary failed:", r.stderr[-400:])


if __name__ == "__main__":
    main()
