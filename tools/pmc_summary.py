#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc CSV run: per kernel, mean counter value per dispatch and the
kernel's mean duration (from the kernel trace of the same run)."""
import csv
import sys
from collections import defaultdict

cc, kt = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
vals = defaultdict(lambda: defaultdict(list))
durs = defaultdict(list)
seen = set()
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].split("(")[0]
    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        durs[k].append(dur.get(r["Dispatch_Id"], 0.0))
for k in vals:
    n = len(durs[k])
    print(f"{k}  dispatches={n} avg_us={sum(durs[k]) / max(n, 1):.1f}")
    for c, v in sorted(vals[k].items()):
        print(f"    {c:28s} {sum(v) / len(v):18.1f}")
