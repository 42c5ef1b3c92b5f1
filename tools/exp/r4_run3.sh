mkdir -p gpurun_out/r4d
for v in ${VARIANTS:-nq_base nq_halfx nq_nocw}; do
  export VSOM_ASM_HSACO=$PWD/tools/exp/bin/$v.hsaco
  timeout -k 10 300 python bench.py --config c2 --no-cpu --no-other-arith --steps 30 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v c2', d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
  VSOM_SIM_STEPS=20 timeout -k 10 300 python tools/rank_sim_bench.py 4 8 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v ranksim', d['world'], d['ms_per_step'], d['kernel_ms']['update'])" || exit 1
done
