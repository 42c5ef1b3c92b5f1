set -x
mkdir -p gpurun_out/r4c
for v in nq_base nq_ct16; do
  if [ -n "$v" ]; then export VSOM_ASM_HSACO=$PWD/tools/exp/bin/$v.hsaco; fi
  timeout -k 10 300 python bench.py --config c2 --no-cpu --no-other-arith --steps 30 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])" >> gpurun_out/r4c/c2.txt 2>&1 || exit 1
  VSOM_SIM_STEPS=20 timeout -k 10 300 python tools/rank_sim_bench.py 4 8 >> gpurun_out/r4c/ranksim_$v.jsonl 2>&1 || exit 1
done
cat gpurun_out/r4c/c2.txt; grep -h update gpurun_out/r4c/ranksim_*.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['world'], d['ms_per_step'], d['kernel_ms']['update'])"
for v in nq_base nq_ct16; do
export VSOM_ASM_HSACO=$PWD/tools/exp/bin/$v.hsaco
(VSOM_UPD_NQ=1 VSOM_ASM_SWEEP_N=24 timeout -k 10 600 python -m pytest tests/test_gpu_random_shapes.py tests/test_gpu_compact.py tests/test_gpu_batch_parity.py -x -q -m gpu 2>&1 | tail -3) || exit 1
done
