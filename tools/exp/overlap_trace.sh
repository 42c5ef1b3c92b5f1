#!/bin/bash
# Development: do the two column kernels of a split update plan overlap in time?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/overlap
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/tools/exp/upd_scaling.py "$@" > $O/b.log 2>&1
kt=$(find $O/b -name '*kernel_trace.csv' | head -1)
python3 - "$kt" <<'PY' > $O/timeline.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "update" in r["Kernel_Name"] or "sigma_fin" in r["Kernel_Name"] or "cwp" in r["Kernel_Name"]]
rows = rows[-12:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r["Kernel_Name"][:34], "queue", r.get("Queue_Id"), "start %.3f ms end %.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
PY
rm -rf $O/b
cat $O/timeline.txt
