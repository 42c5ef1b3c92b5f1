#!/bin/bash
# Development: the records profiles/ holds for a round beside the profile set (tools/prof_all.sh): rank simulation,
# reference-API runs, the reference's own perf-harness scenarios, the online sweep, the default bench line.  ROUND=r6.
cd $GRAFT_REPO_ROOT
R=${ROUND:-r6}
O=gpurun_out/${R}_final
mkdir -p $O
python tools/rank_sim_bench.py 1 2 4 8 > $O/rank_sim_strong.jsonl 2>/dev/null; echo rank sim done
bash tools/e2e_bench.sh $O/e2e.jsonl > /dev/null 2>&1; echo e2e done
python tests/perf/ref_harness.py --scale 100 > $O/ref_harness.jsonl 2> $O/ref_harness.err; echo harness done
[ -n "${SWEEP:-}" ] && { timeout -k 10 500 python tools/online_sweep.py 384 > $O/online_sweep.jsonl 2>/dev/null; echo sweep done; }
python bench.py > $O/bench_default.json 2> /dev/null; echo bench done
