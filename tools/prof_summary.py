"""tools/prof_summary.py -- condense a rocprofv3 --kernel-trace --stats output directory into a small
text summary (top kernels by total time) that is committed under profiles/."""
import csv
import glob
import os
import sys


def main(d, out):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    if not rows:
        # fall back to the raw trace
        agg = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    a = agg.setdefault(r["Kernel_Name"], [0, 0])
                    a[0] += 1
                    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        rows = [{"Name": k, "Calls": v[0], "TotalDurationNs": v[1], "AverageNs": v[1] / v[0]} for k, v in agg.items()]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats summary (%s)\n" % d)
        o.write("%-8s %10s %12s %7s  %s\n" % ("calls", "avg_us", "total_ms", "pct", "kernel"))
        for r in rows[:40]:
            o.write("%-8s %10.2f %12.3f %6.2f%%  %s\n" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6,
                                                      100 * float(r["TotalDurationNs"]) / tot, r["Name"]))
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
