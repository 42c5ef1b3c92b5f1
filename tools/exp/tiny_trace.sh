#!/bin/bash
# Development: kernel trace of the 10 x 10 x 9 scenario's epochs at the C ABI (tools/exp/tiny_epoch_calls.py): which kernels an
# epoch launches, how long each runs, the gaps between them
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tiny_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -o k -- python3 $R/tools/exp/tiny_epoch_calls.py > $O/log.txt 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
m = glob.glob(sys.argv[1] + "/prof/**/*memory_copy_trace.csv", recursive=True)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows]
if m:
    ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")) for r in csv.DictReader(open(m[0]))]
ev.sort()
# the last 12 events before the middle of the run
mid = len(ev) // 3
prev = None
for s, e, n in ev[mid:mid + 16]:
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"gap {gap:7.2f} us   run {(e - s) / 1e3:7.2f} us   {n}")
    prev = e
PY
