#!/usr/bin/env python3
"""Cycle stamps of one sample of the one-launch online chunk (development build: tools/exp/build_dev.sh, VSOM_LIB=...):
phases of wavefront 0 and of the last wavefront, in shader cycles"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd
from vsom_amd import capi
L = capi.lib()
out = (C.c_ulonglong * 64)()
names = ["-", "-", "B distances+min (+ post of the previous sample)", "wait barrier 1", "C key+table", "C update", "D squares", "A of the next sample", "wait barrier 2"]
for (W, H, J, B, sg) in ((10, 10, 9, 20, 8.0), (10, 10, 9, 20, 1.0), (32, 32, 4, 40, 8.0), (16, 16, 16, 40, 8.0)):
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=gen.random_map(W * H, J, seed=1))
    X = gen.blobs(B, J, 4, 1, 2, sigma=0.3)
    for _ in range(3):
        ctx.upload_chunk(X)
        ctx.train_online_chunk(0.01, sg, 0)
    assert L.vsom_dev_tiny_stamps(out) == 0
    t = [int(v) for v in out]
    for w, base in (("wavefront 0", 0), ("last wavefront", 16)):
        d = [t[base + i + 1] - t[base + i] for i in range(2, 9)]
        print(f"{W}x{H}x{J} sigma {sg} {w}: total {t[base + 9] - t[base + 2]} cycles | " + ", ".join(f"{n} {v}" for n, v in zip(names[2:], d)))
    ctx.close()
