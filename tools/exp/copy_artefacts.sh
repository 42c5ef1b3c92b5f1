#!/bin/bash
# Development: copy what tools/exp/final_artefacts.sh left under gpurun_out/ into profiles/ (run here, after the GPU call)
cd "$(dirname "$0")/../.."
for c in c3 c3_sigma c3_contracted c2 c4 c5 online; do
  for f in bench_default.json bench_under_rocprof.json kernel_stats.txt pmc.txt bench_under_rocprof_headline.json kernel_stats_headline.txt; do
    [ -s gpurun_out/r5_$c/summary/$f ] && cp gpurun_out/r5_$c/summary/$f profiles/r5_${c}_$f
  done
done
cp gpurun_out/r5_records/ab_stage.jsonl profiles/r5_ab_stage.jsonl
cp gpurun_out/r5_records/ab_cwl2.jsonl profiles/r5_ab_cwl2.jsonl
cp gpurun_out/r5_records/step_timeline.txt profiles/r5_step_timeline.txt
cp gpurun_out/r5_rank_sim_strong.jsonl profiles/r5_rank_sim_strong.jsonl
cp gpurun_out/r5_e2e_final.jsonl profiles/r5_e2e.jsonl
cp gpurun_out/r5_bench_default_final.json profiles/r5_c3_bench_data_variants.json
python3 tools/traffic_from_pmc.py r5 > /dev/null
