(timeout -k 10 900 python -m pytest tests/test_gpu_shortlist.py tests/test_gpu_baseline_configs.py tests/test_gpu_compact.py tests/test_gpu_batch_parity.py tests/test_gpu_fullsize_properties.py -x -q -m gpu 2>&1 | tail -12) || exit 1
for cfg in c3 c2; do
timeout -k 10 300 python bench.py --config $cfg --no-cpu --no-other-arith --no-data-variants --steps 30 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['bmu_shortlist_last'])" || exit 1
done
