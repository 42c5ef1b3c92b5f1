#!/bin/bash
# Development: PMC counters of the update kernel for a list of (node shard, chunk) shapes.
# usage: upd_traffic.sh "<counters>" shape...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/updtraffic
mkdir -p $O
CTRS=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/tools/exp/upd_scaling.py "$@" > $O/b.log 2>&1
cc=$(find $O/b -name '*counter_collection.csv' | head -1)
kt=$(find $O/b -name '*kernel_trace.csv' | head -1)
python3 - "$cc" "$kt" <<'PY' > $O/per_dispatch.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
agg = collections.OrderedDict()
for r in rows:
    if "update" not in r["Kernel_Name"]:
        continue
    key = (r["Dispatch_Id"], r.get("Grid_Size", ""))
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in agg.items():
    print(k, "ms=%.3f" % dur.get(k[0], -1), {a: "%.4g" % b for a, b in v.items()})
PY
rm -rf $O/b
tail -3 $O/b.log | cut -c1-200
