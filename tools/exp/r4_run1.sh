set -x
mkdir -p gpurun_out/r4a
(VSOM_UPD_NQ=1 VSOM_ASM_SWEEP_N=24 timeout -k 10 600 python -m pytest tests/test_gpu_random_shapes.py tests/test_gpu_compact.py tests/test_gpu_batch_parity.py -x -q -m gpu > gpurun_out/r4a/nq_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r4a/nq_tests.log)
tail -5 gpurun_out/r4a/nq_tests.log
grep -q "rc=0" gpurun_out/r4a/nq_tests.log || exit 1
VSOM_UPD_NQ=0 timeout -k 10 300 python tools/rank_sim_bench.py 1 2 4 8 > gpurun_out/r4a/ranksim_nq0.jsonl 2>&1 &&
timeout -k 10 300 python tools/rank_sim_bench.py 1 2 4 8 > gpurun_out/r4a/ranksim_nq_default.jsonl 2>&1 &&
VSOM_UPD_NQ=0 timeout -k 10 300 python bench.py --config c2 --no-cpu --no-other-arith > gpurun_out/r4a/c2_nq0.json 2>&1 &&
timeout -k 10 300 python bench.py --config c2 --no-cpu --no-other-arith > gpurun_out/r4a/c2_nq.json 2>&1 &&
VSOM_UPD_NQ=1 timeout -k 10 300 python bench.py --config c3 --no-cpu --no-other-arith > gpurun_out/r4a/c3_nq1.json 2>&1
cat gpurun_out/r4a/ranksim_nq0.jsonl gpurun_out/r4a/ranksim_nq_default.jsonl
