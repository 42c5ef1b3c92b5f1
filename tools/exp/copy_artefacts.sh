#!/bin/bash
# Development: copy what the round's GPU calls (tools/prof_all.sh, tools/exp/final_artefacts.sh) left under gpurun_out/ into
# profiles/ (run here, after the GPU calls).  ROUND=r6 by default.
cd "$(dirname "$0")/../.."
R=${ROUND:-r6}
for c in c3 c3_sigma c3_contracted c2 c4 c5 online online_exact; do
  for f in bench_default.json bench_under_rocprof.json kernel_stats.txt pmc.txt bench_under_rocprof_headline.json kernel_stats_headline.txt; do
    [ -s gpurun_out/${R}_$c/summary/$f ] && cp gpurun_out/${R}_$c/summary/$f profiles/${R}_${c}_$f
  done
done
for f in rank_sim_strong.jsonl e2e.jsonl ref_harness.jsonl online_sweep.jsonl; do
  [ -s gpurun_out/${R}_final/$f ] && cp gpurun_out/${R}_final/$f profiles/${R}_$f
done
[ -s gpurun_out/${R}_final/bench_default.json ] && cp gpurun_out/${R}_final/bench_default.json profiles/${R}_c3_bench_data_variants.json
python3 tools/traffic_from_pmc.py $R > /dev/null
