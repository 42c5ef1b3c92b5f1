set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
VSOM_SL_SWEEP_N=200 timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize_properties.py -x -q > gpurun_out/r5g/tests.log 2>&1 || { tail -40 gpurun_out/r5g/tests.log; exit 1; }
tail -2 gpurun_out/r5g/tests.log
bash tools/exp/kstats.sh r5g_c3 --steps 20 | head -12
python bench.py --no-cpu > gpurun_out/r5g/bench_c3.json 2> gpurun_out/r5g/bench_c3.err
python tools/exp/show_bench.py gpurun_out/r5g/bench_c3.json
