set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
python tools/rank_sim_bench.py 1 2 4 8 > gpurun_out/r5m/rank_sim.jsonl 2> gpurun_out/r5m/rank_sim.err
cat gpurun_out/r5m/rank_sim.jsonl
python bench.py --config c4 --no-cpu > gpurun_out/r5m/bench_c4.json 2> gpurun_out/r5m/bench_c4.err
python bench.py --config online --no-cpu > gpurun_out/r5m/bench_online.json 2> gpurun_out/r5m/bench_online.err
python tools/exp/show_bench.py gpurun_out/r5m/bench_c4.json gpurun_out/r5m/bench_online.json
