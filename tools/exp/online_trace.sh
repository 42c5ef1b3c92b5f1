#!/bin/bash
# Development: kernel durations and gaps of the online path (rocprofv3 kernel trace).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/online_trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/tools/online_bench.py 128 784 256 > $O/b.log 2>&1
kt=$(find $O/b -name '*kernel_trace.csv' | head -1)
python3 - "$kt" <<'PY' > $O/timeline.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "online" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# per kernel-name stats + gaps to the previous online kernel, split by run (4 sigmas x 2 chunks x 256 samples)
seg = collections.OrderedDict()
prev_end = None
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-28:]
    grid = r.get("Grid_Size", "")
    key = (name, grid)
    d = seg.setdefault(key, {"n": 0, "dur": 0.0, "gap": 0.0})
    d["n"] += 1
    d["dur"] += (e - s) / 1e3
    if prev_end is not None and s - prev_end < 100000:
        d["gap"] += (s - prev_end) / 1e3
    prev_end = e
for k, d in seg.items():
    print(k, "n=%d avg_dur=%.2f us avg_gap_before=%.2f us" % (d["n"], d["dur"] / d["n"], d["gap"] / d["n"]))
PY
rm -rf $O/b
cat $O/timeline.txt; grep sigma $O/b.log
