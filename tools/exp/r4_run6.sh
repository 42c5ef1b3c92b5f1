(VSOM_ASM_SWEEP_N=40 timeout -k 10 900 python -m pytest tests/test_gpu_random_shapes.py tests/test_gpu_compact.py tests/test_gpu_batch_parity.py tests/test_gpu_goldens.py tests/test_gpu_baseline_configs.py -x -q -m gpu 2>&1 | tail -15) || exit 1
timeout -k 10 300 python tools/configs_bench.py c2median 2>&1 | cut -c1-260 || exit 1
