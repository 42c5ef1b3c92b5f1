#!/bin/bash
# Builds libvsom_hip.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off + correctly rounded div/sqrt: the strict kernels must reproduce the
# reference's SSE2 (non-FMA) fp32 results bit for bit.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${1:-$here/../libvsom_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function"
# hand-scheduled update kernel: generator -> .s -> code object -> C array included by vsom_update.hip
LLVM="${LLVM_BIN:-/opt/rocm/lib/llvm/bin}"
if [ ! -f "$here/vsom_update_hsaco.inc" ] || [ "$here/gen_update_asm.py" -nt "$here/vsom_update_hsaco.inc" ] || [ "$here/gen_nt_asm.py" -nt "$here/vsom_update_hsaco.inc" ]; then
  python3 "$here/gen_update_asm.py" "$here/vsom_update_gfx950.s"
  "$LLVM/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$here/vsom_update_gfx950.s" -o "$here/vsom_update_gfx950.o"
  "$LLVM/ld.lld" -shared "$here/vsom_update_gfx950.o" -o "$here/vsom_update_gfx950.hsaco"
  python3 - "$here/vsom_update_gfx950.hsaco" "$here/vsom_update_hsaco.inc" <<'PY'
import sys
b = open(sys.argv[1], "rb").read()
open(sys.argv[2], "w").write(",".join(str(x) for x in b) + "\n")
PY
  rm -f "$here/vsom_update.o"
fi
objs=()
for f in vsom_capi vsom_bmu vsom_shortlist vsom_update vsom_online vsom_tiny vsom_group vsom_compact vsom_xq vsom_sl_i8; do
  o="$here/$f.o"
  if [ ! -f "$o" ] || [ "$here/$f.hip" -nt "$o" ] || [ "$here/vsom_internal.hpp" -nt "$o" ] || [ "$here/vsom_device.hpp" -nt "$o" ] || [ "$here/vsom_digits.hpp" -nt "$o" ] || [ "$here/../../include/vsom_hip.h" -nt "$o" ]; then
    $HIPCC $FLAGS -c "$here/$f.hip" -o "$o" &
  fi
  objs+=("$o")
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}" -ldl
echo "built $out"
