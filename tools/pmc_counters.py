"""tools/pmc_counters.py -- per-kernel means of rocprofv3 PMC counters: reads every *counter_collection.csv below the given
directories and prints, for the kernels whose name contains one of the filters, the mean of each counter per launch.

    python tools/pmc_counters.py OUT.txt "label=dir1,dir2,..." ... -- filter1 filter2 ...
"""
import csv
import glob
import os
import sys


def collect(dirs):
    acc = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"], r["Counter_Name"])
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def main():
    out = sys.argv[1]
    split = sys.argv.index("--")
    groups = [a.split(":", 1) if ":" in a.split("=")[0] else a.split("=", 1) for a in sys.argv[2:split]]
    filters = sys.argv[split + 1:]
    lines = []
    for label, dirs in groups:
        means, counts = collect(dirs.split(","))
        kernels = sorted({k for k, _ in means if any(f in k for f in filters)})
        for kn in kernels:
            lines.append("[%s] %s" % (label, kn[:150]))
            cs = sorted(c for k, c in means if k == kn)
            for c in cs:
                lines.append("    %-32s %16.1f   (%d launches)" % (c, means[(kn, c)], counts[(kn, c)]))
            m = {c: means[(kn, c)] for c in cs}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m and m["SQ_BUSY_CU_CYCLES"]:
                lines.append("    MFMA busy / CU busy cycles       %16.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CU_CYCLES"]))
            if "SQ_INSTS_VALU" in m and "SQ_INSTS_VALU_MFMA_MOPS_BF16" in m:
                pass
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
