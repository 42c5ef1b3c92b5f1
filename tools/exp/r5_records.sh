#!/bin/bash
# Round-5 records kept under profiles/: interleaved A/B runs and the kernel timeline of two timed C3 steps.
#   bash tools/exp/r5_records.sh   (on the GPU box; needs tools/exp/bin/libvsom_dev.so + nt_base / nt_cwl2 code objects)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_records
mkdir -p $O
cd $R
: > $O/ab_stage.jsonl
for m in 48 64 80 96 112 128; do
  timeout -k 10 200 python tools/exp/ab_stage.py --map $m --timers update 2>/dev/null | grep '^{' >> $O/ab_stage.jsonl
done
: > $O/ab_cwl2.jsonl
VSOM_LIB=$R/tools/exp/bin/libvsom_dev.so timeout -k 10 200 python tools/exp/ab_hsaco.py tools/exp/bin/nt_base.hsaco tools/exp/bin/nt_cwl2.hsaco 2>/dev/null | grep '^{' >> $O/ab_cwl2.jsonl
VSOM_LIB=$R/tools/exp/bin/libvsom_dev.so timeout -k 10 200 python tools/exp/ab_hsaco.py tools/exp/bin/nt_base.hsaco tools/exp/bin/nt_base.hsaco 2>/dev/null | grep '^{' >> $O/ab_cwl2.jsonl
bash tools/exp/step_timeline.sh r5_records_tl > /dev/null 2>&1
cp $R/gpurun_out/r5_records_tl/timeline.txt $O/step_timeline.txt
echo records done
