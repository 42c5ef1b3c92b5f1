set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py tests/test_gpu_switches.py tests/test_gpu_compact.py -x -q > gpurun_out/r5n/tests.log 2>&1 || { tail -40 gpurun_out/r5n/tests.log; exit 1; }
tail -2 gpurun_out/r5n/tests.log
python bench.py --config c4 --no-cpu > gpurun_out/r5n/bench_c4.json 2> gpurun_out/r5n/bench_c4.err
python bench.py --no-cpu --no-data-variants > gpurun_out/r5n/bench_c3.json 2> gpurun_out/r5n/bench_c3.err
python tools/exp/show_bench.py gpurun_out/r5n/bench_c4.json gpurun_out/r5n/bench_c3.json
bash tools/exp/kstats.sh r5n_c4 --config c4 --steps 20
