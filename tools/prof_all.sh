#!/bin/bash
# Profile set of a round: ROUND=r6 bash tools/prof_all.sh [cfg ...] (on the GPU box; default: every configuration); summaries
# land in gpurun_out/<round>_<tag>/summary/ and are copied to profiles/<round>_<tag>_* (tools/exp/copy_artefacts.sh).
# `bench_default.json` of <round>_c3 is the driver's command line.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=${ROUND:-r6}
CFGS="${*:-c3 c3_sigma c3_contracted c2 c4 c5 online}"
for c in $CFGS; do
  case $c in
    c3) bash tools/collect_profiles.sh ${R}_c3 --steps 20 > /dev/null 2>&1 ;;
    c3_sigma) bash tools/collect_profiles.sh ${R}_c3_sigma --steps 10 --arith sigma --no-other-arith --no-data-variants > /dev/null 2>&1 ;;
    c3_contracted) bash tools/collect_profiles.sh ${R}_c3_contracted --steps 10 --arith contracted --no-other-arith --no-data-variants > /dev/null 2>&1 ;;
    c2) bash tools/collect_profiles.sh ${R}_c2 --config c2 --steps 10 > /dev/null 2>&1 ;;
    c4) bash tools/collect_profiles.sh ${R}_c4 --config c4 --steps 10 > /dev/null 2>&1 ;;
    c5) bash tools/collect_profiles.sh ${R}_c5 --config c5 --steps 10 > /dev/null 2>&1 ;;
    online) bash tools/collect_profiles.sh ${R}_online --config online --steps 5 > /dev/null 2>&1
            bash tools/collect_profiles.sh ${R}_online_exact --config online --steps 5 --online-search exact > /dev/null 2>&1 ;;
  esac
  echo $c done
done
