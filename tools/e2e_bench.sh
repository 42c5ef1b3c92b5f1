#!/bin/bash
# End-to-end numbers through the reference API (Som::train via ArrayDataLoader / MnistDataLoader -> DataSet -> the C ABI),
# one JSON line per run into $1 (default gpurun_out/e2e.jsonl).  Rows: tests/gen.py mnist_like (uint8-valued MNIST-like
# pixels), 16384 x 784 for the array loader, an MNIST-sized IDX pair (60000 x 784 + labels) for the reference's loader.
#   usage: tools/e2e_bench.sh [out.jsonl]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$R/gpurun_out/e2e.jsonl}
mkdir -p "$(dirname "$OUT")"
D=$(mktemp -d)
trap 'rm -rf "$D"' EXIT
python3 - "$D" "$R" <<'PY'
import struct, sys, numpy as np
d, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root + "/tests")
import gen
x = gen.mnist_like(16384, seed=3, dim=784)
x.tofile(d + "/rows.f32")
n = 60000
img = np.concatenate([gen.mnist_like(4096, seed=10 + i, dim=784) for i in range(15)])[:n].astype(np.uint8)
lab = np.random.RandomState(3).randint(0, 10, size=n).astype(np.uint8)
open(d + "/train-images-idx3-ubyte", "wb").write(struct.pack(">IIII", 0x803, n, 28, 28) + img.tobytes())
open(d + "/train-labels-idx1-ubyte", "wb").write(struct.pack(">II", 0x801, n) + lab.tobytes())
PY
T=$R/variational-self-organizing-maps_amd/host/host_api_test
: > "$OUT"
for mode in strict sigma; do
    VSOM_UPDATE_MODE=$mode $T perf_e2e array "$D/rows.f32" 16384 784 4096 | grep '^{' >> "$OUT"
    VSOM_UPDATE_MODE=$mode $T perf_e2e mnist "$D" 4096 | grep '^{' >> "$OUT"
done
# the online drivers (Som::train(Exponential | InverseProportional): one trainSingle per sample, Som.cpp:1135-1187)
$T perf_e2e_online "$D/rows.f32" 16384 784 4096 | grep '^{' >> "$OUT"
cat "$OUT"
