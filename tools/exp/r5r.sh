set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py -x -q > gpurun_out/r5r/tests.log 2>&1 || { tail -40 gpurun_out/r5r/tests.log; exit 1; }
tail -2 gpurun_out/r5r/tests.log
python bench.py --no-cpu > gpurun_out/r5r/bench_c3.json 2> gpurun_out/r5r/bench_c3.err
python bench.py --config c2 --no-cpu > gpurun_out/r5r/bench_c2.json 2> gpurun_out/r5r/bench_c2.err
python tools/exp/show_bench.py gpurun_out/r5r/bench_c3.json gpurun_out/r5r/bench_c2.json
