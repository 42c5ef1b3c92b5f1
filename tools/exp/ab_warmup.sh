#!/bin/bash
# Development: does the timed region of the default line still see the clock ramp?  value / staged_in_step at several warm-ups
cd $GRAFT_REPO_ROOT
for w in 3 10 30 3 10 30; do
  r=$(python bench.py --warmup $w --no-cpu --no-other-arith --no-data-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['kernel_ms_per_step']['update'], d['staged_in_step']['ms_per_step'])")
  echo "warmup=$w: $r"
done
