#!/bin/bash
# Commands behind profiles/r1_*: run on the GPU box (gpurun), outputs under gpurun_out/.
# rocprofv3 must start the program itself (python3 ...), from /tmp with TMPDIR=/tmp; PMC passes are
# separate runs without any other trace domain.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p4a -o r -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu
rocprofv3 --pmc FETCH_SIZE TCC_HIT --kernel-trace --output-format csv -d $R/gpurun_out/p4b -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu
rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $R/gpurun_out/p4c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/p4d -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu
# summaries:  python tools/rocpd_summary.py gpurun_out/p4a/r_results.db ; python tools/pmc_summary.py <counter_collection.csv> <kernel_trace.csv>
