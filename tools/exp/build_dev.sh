#!/bin/bash
# development build of the library (-DVSOM_DEVELOPMENT: honours VSOM_ASM_HSACO) -> tools/exp/bin/libvsom_dev.so; load it with
# VSOM_LIB=... (variational-self-organizing-maps_amd/capi.py).  The shipped library is not touched.
set -euo pipefail
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/variational-self-organizing-maps_amd/csrc
mkdir -p $R/tools/exp/bin /tmp/vsom_dev_objs
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -DVSOM_DEVELOPMENT"
for f in vsom_capi vsom_bmu vsom_shortlist vsom_update vsom_online vsom_tiny vsom_group vsom_compact vsom_xq vsom_sl_i8; do
  /opt/rocm/bin/hipcc $FLAGS -c $C/$f.hip -o /tmp/vsom_dev_objs/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/bin/libvsom_dev.so /tmp/vsom_dev_objs/*.o -ldl
echo built $R/tools/exp/bin/libvsom_dev.so
