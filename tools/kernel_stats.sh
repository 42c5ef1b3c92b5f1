#!/bin/bash
# kernel-trace statistics of one bench configuration: bash tools/kernel_stats.sh TAG [bench args]
TAG=${1:-r4_stats}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/a -o r -- python3 $R/bench.py --warmup 3 --no-cpu --no-other-arith --no-data-variants "$@" > $O/bench_under_rocprof.json 2> $O/a.err
python3 $R/tools/rocpd_summary.py $O/a/r_results.db > $O/kernel_stats.txt 2>&1
rm -rf $O/a
cat $O/kernel_stats.txt
