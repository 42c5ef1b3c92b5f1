#!/bin/bash
# Development: the headline step with the next chunk's staging kernels behind the chains (default) against staging at commit
# (VSOM_NO_LATE_AHEAD=1), interleaved on one box
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  for v in 1 0; do
    r=$(VSOM_NO_LATE_AHEAD=$v python bench.py --no-cpu --no-other-arith --no-data-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step']['stage'], d['kernel_ms_per_step']['update'], d['staged_in_step']['ms_per_step'])")
    echo "no_late_ahead=$v: $r"
  done
done
