set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py -x -q -m gpu -k "clr or c5" > gpurun_out/r5f/tests.log 2>&1 || { tail -40 gpurun_out/r5f/tests.log; exit 1; }
tail -2 gpurun_out/r5f/tests.log
STEPS=8 python tools/exp/c5_steps.py shortlist | cut -c1-40,240-420
python bench.py --config c5 --no-cpu > gpurun_out/r5f/bench_c5.json 2> gpurun_out/r5f/bench_c5.err
python tools/exp/show_bench.py gpurun_out/r5f/bench_c5.json
