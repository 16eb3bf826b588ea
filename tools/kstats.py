#!/usr/bin/env python3
"""Print per-kernel count / avg / total (us) from a rocprofv3 --kernel-trace CSV (csv module: names contain commas).
usage: python tools/kstats.py <dir-or-file>"""
import csv, glob, os, sys
from collections import defaultdict
p = sys.argv[1]
files = [p] if os.path.isfile(p) else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
acc = defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0][:70]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        acc[n][0] += 1
        acc[n][1] += d
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %6d calls  avg %9.2f us  total %10.1f us" % (n, c, t / c, t))
