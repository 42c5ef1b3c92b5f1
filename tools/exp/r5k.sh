set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py tests/test_gpu_compact.py tests/test_gpu_fullsize_properties.py tests/test_gpu_group.py tests/test_gpu_dist_ranks.py -x -q -m gpu > gpurun_out/r5k/tests.log 2>&1 || { tail -40 gpurun_out/r5k/tests.log; exit 1; }
tail -2 gpurun_out/r5k/tests.log
python tools/exp/c3_search_time.py 2>/dev/null | tail -1
bash tools/exp/sl_dbg2.sh 2>&1 | grep "dbg=0\|dbg=4"
python bench.py --no-cpu > gpurun_out/r5k/bench_c3.json 2> gpurun_out/r5k/bench_c3.err
python bench.py --config c2 --no-cpu > gpurun_out/r5k/bench_c2.json 2> gpurun_out/r5k/bench_c2.err
python tools/exp/show_bench.py gpurun_out/r5k/bench_c3.json gpurun_out/r5k/bench_c2.json
