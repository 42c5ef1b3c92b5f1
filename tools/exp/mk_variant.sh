#!/bin/bash
# development: build a variant of the hand-scheduled code object (generator env knobs) -> tools/exp/bin/NAME.hsaco,
# to be timed with VSOM_ASM_HSACO=... (development builds of the library only)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/variational-self-organizing-maps_amd/csrc
L=/opt/rocm/lib/llvm/bin
mkdir -p $R/tools/exp/bin
env "$@" python3 $C/gen_update_asm.py /tmp/$name.s
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/$name.s -o /tmp/$name.o
$L/ld.lld -shared /tmp/$name.o -o $R/tools/exp/bin/$name.hsaco
echo built $name
