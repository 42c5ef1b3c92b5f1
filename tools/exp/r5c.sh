set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
bash tools/e2e_bench.sh gpurun_out/r5c/e2e.jsonl > gpurun_out/r5c/e2e.log 2>&1 || { tail -20 gpurun_out/r5c/e2e.log; exit 1; }
python bench.py --no-cpu > gpurun_out/r5c/bench_c3.json 2> gpurun_out/r5c/bench_c3.err || { tail -20 gpurun_out/r5c/bench_c3.err; exit 1; }
python bench.py --config c2 --no-cpu > gpurun_out/r5c/bench_c2.json 2> gpurun_out/r5c/bench_c2.err
echo benches done
python -m pytest tests -x -q -m gpu > gpurun_out/r5c/tests.log 2>&1 || { tail -40 gpurun_out/r5c/tests.log; exit 1; }
tail -3 gpurun_out/r5c/tests.log
