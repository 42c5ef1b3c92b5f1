#!/bin/bash
# Builds the host-side C++ mirror (libsom_hip.so) and its test program against libvsom_hip.so.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
CXX="${CXX:-g++}"
FLAGS="-std=c++20 -O2 -fPIC -Wall -Wextra -Wno-unused-parameter -I$here/include -I$here/../../include"
$CXX $FLAGS -shared "$here/src/vsom_host.cpp" "$here/src/vsom_custom.cpp" "$here/src/vsom_loaders.cpp" "$here/src/vsom_checkpoint.cpp" -o "$here/libsom_hip.so" -L"$here/.." -lvsom_hip -ldl -pthread -Wl,-rpath,'$ORIGIN/..'
$CXX $FLAGS "$here/tests/host_api_test.cpp" -o "$here/host_api_test" -L"$here" -lsom_hip -L"$here/.." -lvsom_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/..'
$CXX $FLAGS "$here/tests/host_loader_test.cpp" -o "$here/host_loader_test" -L"$here" -lsom_hip -L"$here/.." -lvsom_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/..'
$CXX $FLAGS "$here/tests/host_custom_test.cpp" -o "$here/host_custom_test" -L"$here" -lsom_hip -L"$here/.." -lvsom_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/..'
echo "built $here/libsom_hip.so and host_api_test"
