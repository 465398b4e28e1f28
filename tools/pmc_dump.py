"""tools/pmc_dump.py -- print per-kernel means of every counter found in rocprofv3 counter_collection CSVs under the
given directories (tuning aid).  usage: pmc_dump.py <kernel-substring> dir [dir ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

sub = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                key = r["Kernel_Name"][:70] + " grid=" + r.get("Grid_Size", "?")
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
