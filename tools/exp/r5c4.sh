set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c4
timeout -k 10 900 python -m pytest tests/test_gpu_batch_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_random_shapes.py tests/test_gpu_short_vectors.py tests/test_gpu_goldens.py tests/test_gpu_group.py -x -q > gpurun_out/r5c4/tests.log 2>&1 || { tail -40 gpurun_out/r5c4/tests.log; exit 1; }
tail -2 gpurun_out/r5c4/tests.log
bash tools/exp/kstats.sh r5c4_c4 --config c4 --steps 20 | head -4
bash tools/exp/kstats.sh r5c4_c3 --steps 10 | grep cwp
python bench.py --config c4 --no-cpu > gpurun_out/r5c4/bench_c4.json 2> gpurun_out/r5c4/bench_c4.err
python tools/exp/show_bench.py gpurun_out/r5c4/bench_c4.json
