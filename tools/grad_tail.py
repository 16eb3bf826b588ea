"""Timeline of the tail of one value-and-gradient rollout from a rocprofv3 --kernel-trace CSV: everything behind the last sweep
(start offsets and durations in us).   usage: python tools/grad_tail.py <trace dir>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = [i for i, r in enumerate(rows) if "k_rev_chain" in r["Kernel_Name"]][-1]
first = max(i for i in range(last) if "k_mm_bwd_pair" in rows[i]["Kernel_Name"])
t0 = int(rows[first]["End_Timestamp"])
for r in rows[first:last + 1]:
    print("%8.1f %8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][-44:]))
print("tail: %.1f us from the last sweep's end to the chain's end" % ((int(rows[last]["End_Timestamp"]) - t0) / 1e3))
