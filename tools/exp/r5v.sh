set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5v
timeout -k 10 900 python -m pytest tests/test_gpu_compact.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize_properties.py tests/test_gpu_batch_parity.py tests/test_gpu_random_shapes.py tests/test_gpu_fma_mode.py tests/test_gpu_group.py tests/test_gpu_dist_ranks.py tests/test_gpu_late_epochs.py tests/test_gpu_ingest.py tests/test_gpu_switches.py -x -q > gpurun_out/r5v/tests.log 2>&1 || { tail -40 gpurun_out/r5v/tests.log; exit 1; }
tail -2 gpurun_out/r5v/tests.log
bash tools/exp/kstats.sh r5v_c3 --steps 20 | head -9
python bench.py --no-cpu --no-data-variants > gpurun_out/r5v/bench_c3.json 2> gpurun_out/r5v/bench_c3.err
python bench.py --config c2 --no-cpu --no-data-variants > gpurun_out/r5v/bench_c2.json 2> gpurun_out/r5v/bench_c2.err
python tools/exp/show_bench.py gpurun_out/r5v/bench_c3.json gpurun_out/r5v/bench_c2.json
