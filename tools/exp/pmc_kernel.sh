#!/bin/bash
# Development: SQ counters of one kernel of a bench.py run.  usage: pmc_kernel.sh <tag> <kernel name substring> [bench.py arguments]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; K=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/d -o p -- python3 $R/bench.py --no-cpu --no-other-arith --no-data-variants --steps 3 --warmup 1 "$@" > $O/d.log 2>&1
python3 $R/tools/pmc_summary.py $(find $O/d -name '*counter_collection.csv' | head -1) $(find $O/d -name '*kernel_trace.csv' | head -1) > $O/pmc_summary.txt
grep -A14 "$K" $O/pmc_summary.txt | head -40
rm -rf $O/d
