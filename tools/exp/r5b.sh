set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py tests/test_gpu_compact.py tests/test_gpu_fullsize_properties.py -x -q -m gpu > gpurun_out/r5b/tests.log 2>&1 || { tail -40 gpurun_out/r5b/tests.log; exit 1; }
tail -3 gpurun_out/r5b/tests.log
python bench.py --no-cpu > gpurun_out/r5b/bench_c3.json 2> gpurun_out/r5b/bench_c3.err || { tail -20 gpurun_out/r5b/bench_c3.err; exit 1; }
python bench.py --config c2 --no-cpu > gpurun_out/r5b/bench_c2.json 2> gpurun_out/r5b/bench_c2.err
echo done
