set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5t
VSOM_SL_SWEEP_N=300 timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py tests/test_gpu_compact.py tests/test_gpu_group.py tests/test_gpu_dist_ranks.py tests/test_gpu_ingest.py -x -q > gpurun_out/r5t/tests.log 2>&1 || { tail -40 gpurun_out/r5t/tests.log; exit 1; }
tail -2 gpurun_out/r5t/tests.log
bash tools/exp/kstats.sh r5t_c3 --steps 20 | head -12
python bench.py --no-cpu --no-data-variants > gpurun_out/r5t/bench_c3.json 2> gpurun_out/r5t/bench_c3.err
python bench.py --config c2 --no-cpu --no-data-variants > gpurun_out/r5t/bench_c2.json 2> gpurun_out/r5t/bench_c2.err
python tools/exp/show_bench.py gpurun_out/r5t/bench_c3.json gpurun_out/r5t/bench_c2.json
