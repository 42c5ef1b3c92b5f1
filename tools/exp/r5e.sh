set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
python bench.py --no-cpu > gpurun_out/r5e/bench_c3.json 2> gpurun_out/r5e/bench_c3.err || { tail -20 gpurun_out/r5e/bench_c3.err; exit 1; }
python bench.py --no-cpu --no-stage-ahead --no-data-variants --no-other-arith > gpurun_out/r5e/bench_c3_nostage.json 2> gpurun_out/r5e/bench_c3_nostage.err
python bench.py --config c2 --no-cpu > gpurun_out/r5e/bench_c2.json 2> gpurun_out/r5e/bench_c2.err
python bench.py --config c4 --no-cpu > gpurun_out/r5e/bench_c4.json 2> gpurun_out/r5e/bench_c4.err
python bench.py --config c5 --no-cpu > gpurun_out/r5e/bench_c5.json 2> gpurun_out/r5e/bench_c5.err
python bench.py --config online --no-cpu > gpurun_out/r5e/bench_online.json 2> gpurun_out/r5e/bench_online.err
python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu > gpurun_out/r5e/tests.log 2>&1 || { tail -40 gpurun_out/r5e/tests.log; exit 1; }
echo done
