#!/bin/bash
# Commands behind profiles/r<N>_*: run on the GPU box (gpurun), raw outputs under gpurun_out/<tag>/,
# summaries under gpurun_out/<tag>/summary/ (copy the ones to keep into profiles/).
#   tools/collect_profiles.sh TAG [bench.py arguments, e.g. --config c5 --steps 5]
# rocprofv3 must start the program itself (python3 ...), from /tmp with TMPDIR=/tmp; PMC passes are
# separate runs without any other trace domain (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-prof}
shift || true
BARGS="$*"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O/summary
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $O/a -o r -- python3 $R/bench.py --warmup 3 --no-cpu $BARGS > $O/summary/bench_under_rocprof.json 2> $O/a.err
python3 $R/tools/rocpd_summary.py $O/a/r_results.db > $O/summary/kernel_stats.txt 2>&1
echo "kernel stats done"
# the same without the data variants: they run the SAME kernels on other data (float_dense: 5.4 instead of 4.1 ms per chain
# launch), so only here does a kernel's average duration compare with the bench line's `roofline` (timed region = headline data)
timeout -k 10 420 rocprofv3 --kernel-trace --stats -d $O/a2 -o r -- python3 $R/bench.py --warmup 3 --no-cpu --no-data-variants $BARGS > $O/summary/bench_under_rocprof_headline.json 2> $O/a2.err
python3 $R/tools/rocpd_summary.py $O/a2/r_results.db > $O/summary/kernel_stats_headline.txt 2>&1
echo "headline kernel stats done"
timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE TCC_HIT --kernel-trace --output-format csv -d $O/b -o p -- python3 $R/bench.py --no-cpu --no-other-arith --no-data-variants $BARGS --steps 3 --warmup 1 > $O/b.log 2>&1
echo "pmc pass 1 done"
timeout -k 10 420 rocprofv3 --pmc WRITE_SIZE TCC_MISS TCC_REQ --kernel-trace --output-format csv -d $O/c -o p -- python3 $R/bench.py --no-cpu --no-other-arith --no-data-variants $BARGS --steps 3 --warmup 1 > $O/c.log 2>&1
echo "pmc pass 2 done"
timeout -k 10 420 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/d -o p -- python3 $R/bench.py --no-cpu --no-other-arith --no-data-variants $BARGS --steps 3 --warmup 1 > $O/d.log 2>&1
echo "pmc pass 3 done"
for p in b c d; do
  cc=$(find $O/$p -name '*counter_collection.csv' | head -1)
  kt=$(find $O/$p -name '*kernel_trace.csv' | head -1)
  if [ -n "$cc" ] && [ -n "$kt" ]; then
    echo "## pass $p" >> $O/summary/pmc.txt
    python3 $R/tools/pmc_summary.py $cc $kt >> $O/summary/pmc.txt 2>&1
  fi
done
# keep only the summaries in the merge-back (raw traces are large)
rm -rf $O/a $O/a2 $O/b $O/c $O/d
cd $R && python3 bench.py $BARGS > $O/summary/bench_default.json 2> $O/bench_default.err
echo "bench done"
