set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
python tools/rank_sim_bench.py 1 2 4 8 > gpurun_out/r5p/rank_sim.jsonl 2> gpurun_out/r5p/rank_sim.err
cat gpurun_out/r5p/rank_sim.jsonl
bash tools/exp/pmc_kernel.sh r5p_pmc sl_k64 --config c4
