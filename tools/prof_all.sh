#!/bin/bash
# Profile set of a round: bash tools/prof_all.sh (on the GPU box); summaries land in gpurun_out/r5_<tag>/summary/
# and are copied to profiles/r5_<tag>_*.  `bench_default.json` of r5_c3 is the driver's command line.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/collect_profiles.sh r5_c3 --steps 20 > /dev/null 2>&1; echo c3 done
bash tools/collect_profiles.sh r5_c3_sigma --steps 10 --arith sigma --no-other-arith --no-data-variants > /dev/null 2>&1; echo c3 sigma done
bash tools/collect_profiles.sh r5_c3_contracted --steps 10 --arith contracted --no-other-arith --no-data-variants > /dev/null 2>&1; echo c3 contracted done
bash tools/collect_profiles.sh r5_c2 --config c2 --steps 10 > /dev/null 2>&1; echo c2 done
bash tools/collect_profiles.sh r5_c4 --config c4 --steps 10 > /dev/null 2>&1; echo c4 done
bash tools/collect_profiles.sh r5_c5 --config c5 --steps 10 > /dev/null 2>&1; echo c5 done
bash tools/collect_profiles.sh r5_online --config online --steps 5 > /dev/null 2>&1; echo online done
