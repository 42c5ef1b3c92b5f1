#!/usr/bin/env python3
"""Phase stamps of the image-bounded online search (development build only: tools/exp/build_dev.sh, VSOM_LIB=...):
trains chunks on BASELINE's 128 x 128 x 784 map and prints, for the stamped workgroups of the LAST sample pair, the time
between phases in microseconds (100 MHz wall clock)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import gen  # noqa: E402
import vsom_amd  # noqa: E402
from vsom_amd import capi  # noqa: E402

W = H = 128
J = 784
ctx = vsom_amd.Context(W, H, J)
ctx.set_bmu_mode(capi.BMU_SHORTLIST)
ctx.set_state(map=(gen.random_map(W * H, J, seed=42) * np.float32(100) + np.float32(100)).astype(np.float32))
L = capi.lib()
out = (C.c_ulonglong * 32)()
import time
for rep in range(6):
    X = gen.mnist_like(301 + 17 * rep, seed=3 + rep, dim=J)
    ctx.upload_chunk(X)
    t0 = time.perf_counter()
    ctx.train_online_chunk(0.1, 8.0, capi.EXPONENTIAL)
    print(f"  chunk of {X.shape[0]}: {(time.perf_counter() - t0) / X.shape[0] * 1e6:.2f} us per sample (wall, incl. the per-chunk passes)")
    assert L.vsom_dev_onl_stamps(out) == 0
    t = [int(v) for v in out]

    def d(a, b):
        return (t[b] - t[a]) / 100.0
    print(f"rep {rep}: refine (block 0) in->compacted {d(0, 2):.2f}  distances {d(2, 3):.2f}  ->atomic {d(3, 4):.2f} | "
          f"post {d(5, 6):.2f} | window entry->resolved {d(8, 9):.2f}  update {d(9, 10):.2f}  digit+reduce {d(10, 11):.2f}  interval+atomic {d(11, 12):.2f} | "
          f"scan entry->loads {d(16, 17):.2f}  passes {d(17, 18):.2f} | fused: first block -> last block entry {d(20, 21):.2f}  -> last block end {d(20, 22):.2f}",
          flush=True)
tr = (C.c_ulonglong * 8192)()
assert L.vsom_dev_onl_trace(tr) == 0
tr = np.array(tr, dtype=np.uint64).reshape(4096, 2).astype(np.int64)
used = tr[:, 0] > 0
ent, ext = tr[used, 0], tr[used, 1]
t0 = ent.min()
nb = int(used.sum())
print(f"fused launch trace: {nb} workgroups; entry of block b relative to the first entry (us):")
for b in (0, 64, 128, 255, 256, 300, 512, 768, 1024, 1280, 1536, 1792, nb - 1):
    if b < nb:
        print(f"  block {b:5d}: entry {(ent[b] - t0) / 100:.2f}  exit {(ext[b] - t0) / 100:.2f}")
print(f"  last exit {(ext.max() - t0) / 100:.2f}; concurrently resident at t = 1, 2, 3, 4, 5, 6, 7, 8 us:",
      [int(((ent - t0) <= 100 * t).sum() - ((ext - t0) <= 100 * t).sum()) for t in range(1, 9)])
print(f"  boundary fused(B-2) -> refine(B-1): refine block 0 enters {(t[0] - ext.max()) / 100:.2f} us after the fused launch's last exit; "
      f"refine block 0: entry -> atomic {(t[4] - t[0]) / 100:.2f}; window-only fused(B-1) centre workgroup enters {(t[8] - t[4]) / 100:.2f} us after that atomic")
print(ctx.online_search_stats())
ctx.close()
