#!/usr/bin/env python3
"""The launches of the LAST factorisation in a rocprofv3 --kernel-trace CSV of tools/fact_bench.py, in order: duration and the
gap to the previous launch's end (us).  usage: python tools/fact_chain.py <trace dir>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_gram" in r["Kernel_Name"])
prev_end, t0, out = None, int(rows[last]["Start_Timestamp"]), []
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("pilco::", "").replace("void ", "").split("(")[0]
    out.append("%8.2f %-34s %7.2f  gap %6.2f  grid %s" % ((s - t0) / 1e3, n[:34], (e - s) / 1e3, 0.0 if prev_end is None else (s - prev_end) / 1e3, r.get("Grid_Size_X", "") + "x" + r.get("Grid_Size_Y", "") + "x" + r.get("Grid_Size_Z", "")))
    prev_end = e
print("\n".join(out))
print("total %.2f us" % ((prev_end - t0) / 1e3))
