set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5l/tests.log 2>&1 || { tail -40 gpurun_out/r5l/tests.log; exit 1; }
tail -2 gpurun_out/r5l/tests.log
python bench.py --no-cpu > gpurun_out/r5l/bench_c3.json 2> gpurun_out/r5l/bench_c3.err
python bench.py --config c2 --no-cpu > gpurun_out/r5l/bench_c2.json 2> gpurun_out/r5l/bench_c2.err
python bench.py --config c5 --no-cpu > gpurun_out/r5l/bench_c5.json 2> gpurun_out/r5l/bench_c5.err
python tools/exp/show_bench.py gpurun_out/r5l/bench_c3.json gpurun_out/r5l/bench_c2.json gpurun_out/r5l/bench_c5.json
