#!/bin/bash
# Development: the kernel timeline (start / end, queue) of the last two steps of a bench run: which kernels overlap.
#   bash tools/exp/step_timeline.sh TAG [bench args]
set -u
TAG=${1:-timeline}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/bench.py --no-cpu --no-other-arith --no-data-variants --steps 6 --warmup 2 "$@" > $O/b.log 2>&1
kt=$(find $O/b -name '*kernel_trace.csv' | head -1)
python3 - "$kt" <<'PY' > $O/timeline.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# two steps of the TIMED region (dominant-group events only): the third and fourth timed step
starts = [i for i, r in enumerate(rows) if "sl_prepare" in r["Kernel_Name"] or "clr_node_feat" in r["Kernel_Name"]]
i0 = starts[4] - 6 if len(starts) >= 8 else max(0, len(rows) - 40)
rows = rows[:starts[6] + 8] if len(starts) >= 8 else rows
rows = rows[max(i0, 0):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%-44s q%-3s %9.3f -> %9.3f us  (%8.3f)" % (r["Kernel_Name"][:44], r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $O/b
cat $O/timeline.txt
