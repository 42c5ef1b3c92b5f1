#!/bin/bash
# End-to-end C++ path on an MNIST-sized IDX pair (synthetic pixels): loader + DataSet + pipelined training.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$(mktemp -d)
python3 - "$D" <<'PY'
import struct, sys, numpy as np
d = sys.argv[1]
rs = np.random.RandomState(3)
n = 60000
# like tests/gen.py mnist_like: ~19 % non-zero pixels, all inside the central 20x20 window
img = np.zeros((n, 28, 28), np.uint8)
img[:, 4:24, 4:24] = (rs.randint(1, 256, size=(n, 20, 20)) * (rs.rand(n, 20, 20) < 0.19 * 784 / 400)).astype(np.uint8)
lab = rs.randint(0, 10, size=n).astype(np.uint8)
open(d + "/train-images-idx3-ubyte", "wb").write(struct.pack(">IIII", 0x803, n, 28, 28) + img.tobytes())
open(d + "/train-labels-idx1-ubyte", "wb").write(struct.pack(">II", 0x801, n) + lab.tobytes())
PY
$R/variational-self-organizing-maps_amd/host/host_api_test perf_mnist "$D"
rm -rf "$D"
