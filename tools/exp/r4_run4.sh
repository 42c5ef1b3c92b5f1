mkdir -p gpurun_out/r4e
(VSOM_UPD_NT=1 VSOM_ASM_SWEEP_N=24 timeout -k 10 600 python -m pytest tests/test_gpu_random_shapes.py tests/test_gpu_compact.py tests/test_gpu_batch_parity.py -x -q -m gpu 2>&1 | tail -15) || exit 1
for nt in 0 1; do
  export VSOM_UPD_NT=$nt
  for cfg in c2 c3; do
  timeout -k 10 300 python bench.py --config $cfg --no-cpu --no-other-arith --steps 30 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nt=$nt $cfg', d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
  done
  VSOM_SIM_STEPS=20 timeout -k 10 300 python tools/rank_sim_bench.py 2 4 8 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('nt=$nt ranksim', d['world'], d['ms_per_step'], d['kernel_ms']['update'])" || exit 1
done
