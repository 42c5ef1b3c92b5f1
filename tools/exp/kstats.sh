#!/bin/bash
# Development: per-kernel average durations of a bench.py run (rocprofv3 kernel trace).  usage: kstats.sh <tag> [bench.py arguments]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- python3 $R/bench.py --no-cpu --no-other-arith --no-data-variants "$@" > $O/prof.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
out = open(sys.argv[1] + "/kernel_stats.txt", "w")
for r in rows[:28]:
    line = "%-72s %6s %10.1f us %6s%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"])
    print(line); out.write(line + "\n")
PY
