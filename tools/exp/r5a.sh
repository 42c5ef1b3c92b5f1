set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_shortlist.py -x -q -m gpu > gpurun_out/r5a/tests.log 2>&1 || { tail -30 gpurun_out/r5a/tests.log; exit 1; }
tail -3 gpurun_out/r5a/tests.log
python bench.py > gpurun_out/r5a/bench_c3.json 2> gpurun_out/r5a/bench_c3.err || { tail -20 gpurun_out/r5a/bench_c3.err; exit 1; }
python bench.py --config c2 > gpurun_out/r5a/bench_c2.json 2> gpurun_out/r5a/bench_c2.err
bash tools/e2e_bench.sh gpurun_out/r5a/e2e.jsonl > gpurun_out/r5a/e2e.log 2>&1 || { tail -20 gpurun_out/r5a/e2e.log; exit 1; }
echo done
