# Development: everything profiles/ holds for a round, in one GPU call (profile set, A/B records, rank simulation, reference-API run, default bench line).
cd $GRAFT_REPO_ROOT
bash tools/prof_all.sh
bash tools/exp/r5_records.sh
python tools/rank_sim_bench.py 1 2 4 8 > gpurun_out/r5_rank_sim_strong.jsonl 2>/dev/null; echo rank sim done
bash tools/e2e_bench.sh gpurun_out/r5_e2e_final.jsonl > /dev/null 2>&1; echo e2e done
python bench.py > gpurun_out/r5_bench_default_final.json 2> /dev/null; echo bench done
