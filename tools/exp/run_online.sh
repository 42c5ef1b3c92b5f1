(timeout -k 10 900 python -m pytest tests/test_gpu_online.py tests/test_gpu_goldens.py tests/test_gpu_random_shapes.py tests/test_gpu_next_rows.py -x -q -m gpu 2>&1 | tail -8) || exit 1
timeout -k 10 300 python tools/configs_bench.py online 2>&1 | cut -c1-250 || exit 1
timeout -k 10 300 python bench.py --config online --no-cpu | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])"
