mkdir -p gpurun_out/r4f
for nt in 0 1; do
  export VSOM_UPD_NT=$nt
  for ar in sigma contracted; do
  timeout -k 10 300 python bench.py --config c3 --arith $ar --no-cpu --no-other-arith --steps 20 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nt=$nt c3 $ar', d['ms_per_step'], d['roofline']['avg_launch_ms'])" || exit 1
  done
  timeout -k 10 300 python tools/configs_bench.py c2median c3local 2>&1 | cut -c1-260 || exit 1
done
unset VSOM_UPD_NT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
