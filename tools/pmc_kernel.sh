#!/bin/bash
# PMC passes for the dominant update kernel of a bench configuration: bash tools/pmc_kernel.sh TAG [bench args]
set -u
TAG=${1:-r4_c2}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu --no-other-arith --no-data-variants --steps 3 --warmup 1 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/d -o p -- python3 $R/bench.py $B > $O/d.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/e -o p -- python3 $R/bench.py $B > $O/e.log 2>&1
echo skip > $O/f.log
for p in d e f; do
  cc=$(find $O/$p -name '*counter_collection.csv' | head -1)
  kt=$(find $O/$p -name '*kernel_trace.csv' | head -1)
  if [ -n "$cc" ] && [ -n "$kt" ]; then
    echo "## pass $p" >> $O/pmc.txt
    python3 $R/tools/pmc_summary.py $cc $kt | grep -A12 "vsom_update" >> $O/pmc.txt 2>&1
  else
    echo "## pass $p: no output" >> $O/pmc.txt; tail -5 $O/$p.log >> $O/pmc.txt
  fi
done
rm -rf $O/d $O/e $O/f
cat $O/pmc.txt
