set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py tests/test_gpu_compact.py tests/test_gpu_group.py -x -q > gpurun_out/r5s/tests.log 2>&1 || { tail -40 gpurun_out/r5s/tests.log; exit 1; }
tail -2 gpurun_out/r5s/tests.log
bash tools/exp/kstats.sh r5s_c3 --steps 20
python bench.py --no-cpu --no-data-variants > gpurun_out/r5s/bench_c3.json 2> gpurun_out/r5s/bench_c3.err
python bench.py --config c2 --no-cpu --no-data-variants > gpurun_out/r5s/bench_c2.json 2> gpurun_out/r5s/bench_c2.err
python tools/exp/show_bench.py gpurun_out/r5s/bench_c3.json gpurun_out/r5s/bench_c2.json
python tools/rank_sim_bench.py 8 | tee gpurun_out/r5s/rank8.jsonl
