#!/bin/bash
# Development: board power and shader clock while the strict C3 step runs back to back (is the chip power-limited
# under the phase-2 chain kernel?).  Samples rocm-smi twice a second beside a long bench.py run.
#   bash tools/exp/power_sample.sh [strict|sigma|contracted]  > profiles/r3_power_<arith>.txt
ARITH=${1:-strict}
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/bench.py --steps 2500 --warmup 5 --arith $ARITH --no-other-arith --no-data-variants --no-cpu > /tmp/power_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 16); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)\)" | tr '\n' ' ' | sed 's/  */ /g'
  echo
  sleep 0.5
done
wait $BP
python3 -c "
import json; d=json.loads(open('/tmp/power_bench.json').read().strip().splitlines()[-1]); print('bench', d['update_arithmetic'][:16], d['value'], 'samples/s', d['ms_per_step'], 'ms/step', d['kernel_ms_per_step'])"
