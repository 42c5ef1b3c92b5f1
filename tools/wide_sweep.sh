#!/bin/bash
# Wide validation sweeps (GPU box): random shapes against the oracle with several seeds, random groups, the determinism
# soak.  Prints one summary line per run; the log goes to gpurun_out/<tag>/.  usage: bash tools/wide_sweep.sh [tag] [seeds...]
set -u
TAG=${1:-sweep}; shift || true
SEEDS=${*:-"11 12 13"}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$TAG; mkdir -p $O
for s in $SEEDS; do
  (VSOM_SWEEP_N=300 VSOM_ASM_SWEEP_N=150 VSOM_SWEEP_WIDE=1 VSOM_SWEEP_SEED=$s timeout -k 10 900 python -m pytest tests/test_gpu_random_shapes.py -x -q -m gpu 2>&1 | tail -2) > $O/seed_$s.log 2>&1
  echo "seed $s: $(tail -1 $O/seed_$s.log)"
  grep -q "passed" $O/seed_$s.log || exit 1
done
(VSOM_GROUP_SWEEP_N=120 timeout -k 10 900 python -m pytest tests/test_gpu_group.py -x -q -m gpu 2>&1 | tail -2) > $O/groups.log 2>&1
echo "groups: $(tail -1 $O/groups.log)"
grep -q "passed" $O/groups.log || exit 1
(timeout -k 10 600 python tools/soak_determinism.py 2>&1 | tail -3) > $O/soak.log 2>&1
echo "soak: $(tail -1 $O/soak.log)"
