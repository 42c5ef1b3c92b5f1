set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
python -m pytest tests/test_gpu_ingest.py tests/test_gpu_baseline_configs.py tests/test_gpu_bench_contract.py tests/test_gpu_host_cpp.py tests/test_gpu_group.py -x -q -m gpu > gpurun_out/r5d/tests.log 2>&1 || { tail -40 gpurun_out/r5d/tests.log; exit 1; }
tail -3 gpurun_out/r5d/tests.log
python bench.py --no-cpu > gpurun_out/r5d/bench_c3.json 2> gpurun_out/r5d/bench_c3.err || { tail -20 gpurun_out/r5d/bench_c3.err; exit 1; }
python bench.py --no-cpu --no-stage-ahead --no-data-variants --no-other-arith > gpurun_out/r5d/bench_c3_nostage.json 2> gpurun_out/r5d/bench_c3_nostage.err
python bench.py --config c2 --no-cpu > gpurun_out/r5d/bench_c2.json 2> gpurun_out/r5d/bench_c2.err
bash tools/e2e_bench.sh gpurun_out/r5d/e2e.jsonl > gpurun_out/r5d/e2e.log 2>&1 || { tail -20 gpurun_out/r5d/e2e.log; exit 1; }
echo done
