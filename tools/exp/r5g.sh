set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout -k 10 600 python -m pytest tests/test_gpu_online.py -x -q -m gpu > gpurun_out/r5g/tests.log 2>&1 || { tail -40 gpurun_out/r5g/tests.log; exit 1; }
tail -2 gpurun_out/r5g/tests.log
timeout -k 10 300 python bench.py --config online --no-cpu > gpurun_out/r5g/bench_online.json 2> gpurun_out/r5g/bench_online.err
VSOM_NO_LOOKAHEAD=1 timeout -k 10 300 python bench.py --config online --no-cpu > gpurun_out/r5g/bench_online_nola.json 2> gpurun_out/r5g/bench_online_nola.err
python tools/exp/show_bench.py gpurun_out/r5g/bench_online.json gpurun_out/r5g/bench_online_nola.json
