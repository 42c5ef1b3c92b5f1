#!/bin/bash
# Builds libvsom_hip.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off + correctly rounded div/sqrt: the strict kernels must reproduce the
# reference's SSE2 (non-FMA) fp32 results bit for bit.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${1:-$here/../libvsom_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function"
objs=()
for f in vsom_capi vsom_bmu vsom_shortlist vsom_update vsom_online; do
  o="$here/$f.o"
  if [ ! -f "$o" ] || [ "$here/$f.hip" -nt "$o" ] || [ "$here/vsom_internal.hpp" -nt "$o" ] || [ "$here/vsom_device.hpp" -nt "$o" ] || [ "$here/../../include/vsom_hip.h" -nt "$o" ]; then
    $HIPCC $FLAGS -c "$here/$f.hip" -o "$o" &
  fi
  objs+=("$o")
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
