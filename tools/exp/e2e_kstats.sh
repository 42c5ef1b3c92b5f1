#!/bin/bash
# Development: per-kernel durations of the reference-API MNIST path (host_api_test perf_e2e mnist) under rocprofv3.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/e2e_kstats; mkdir -p $O
D=$(mktemp -d); trap 'rm -rf "$D"' EXIT
python3 - "$D" "$R" <<'PY'
import struct, sys, numpy as np
d, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root + "/tests")
import gen
n = 60000
img = np.concatenate([gen.mnist_like(4096, seed=10 + i, dim=784) for i in range(15)])[:n].astype(np.uint8)
lab = np.random.RandomState(3).randint(0, 10, size=n).astype(np.uint8)
open(d + "/train-images-idx3-ubyte", "wb").write(struct.pack(">IIII", 0x803, n, 28, 28) + img.tobytes())
open(d + "/train-labels-idx1-ubyte", "wb").write(struct.pack(">II", 0x801, n) + lab.tobytes())
PY
T=$R/variational-self-organizing-maps_amd/host/host_api_test
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o k -- $T perf_e2e mnist "$D" 4096 > $O/run.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:24]:
    print("%-70s %6s %10.1f us %6s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
