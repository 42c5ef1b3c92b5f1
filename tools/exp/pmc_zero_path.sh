#!/bin/bash
# Development: memory-side counters of the strict chain kernel with and without the zero-slice form (drift between
# the wavefronts of a node group costs L2 reuse of the (c,w) stream).  Counter sets as tools/collect_profiles.sh
# uses them (FETCH_SIZE + TCC_HIT in one pass, TCC_MISS + TCC_REQ in another).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for tag in on off; do
  if [ $tag = off ]; then export VSOM_NO_ZERO_PATH=1; fi
  for pass in a b; do
    if [ $pass = a ]; then CTR="FETCH_SIZE TCC_HIT"; else CTR="WRITE_SIZE TCC_MISS TCC_REQ"; fi
    echo "pass $tag $pass"
    VSOM_EXP_STEPS=3 timeout -k 10 200 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $R/gpurun_out/zp_$tag$pass -o p -- python3 $R/tools/exp/compact_power.py strokes > /dev/null 2>&1 || exit 1
    cc=$(find $R/gpurun_out/zp_$tag$pass -name "*counter_collection.csv" | head -1); kt=$(find $R/gpurun_out/zp_$tag$pass -name "*kernel_trace.csv" | head -1)
    python3 $R/tools/pmc_summary.py $cc $kt >> $R/gpurun_out/r3_zp_$tag.pmc.txt 2>&1; rm -rf $R/gpurun_out/zp_$tag$pass
  done
  echo "== zero path $tag"; grep -A4 "vsom_update_std_rd14" $R/gpurun_out/r3_zp_$tag.pmc.txt | head -12
done
