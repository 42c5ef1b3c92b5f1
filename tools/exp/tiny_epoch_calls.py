#!/usr/bin/env python3
"""where an epoch of the reference's own training scenario (10 x 10 x 9, 20 rows, perf_tests.cpp:74-112) spends its time at
the C ABI: the five calls the C++ mirror's trainBasicSom makes per chunk, timed one by one from ctypes (1-2 us of overhead
each), mean over 300 epochs"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd
from vsom_amd import capi
L = capi.lib()
W, H, J, B = 10, 10, 9, 20
X = np.ascontiguousarray(gen.blobs(B, J, 4, 1, 2, sigma=0.3))
ctx = vsom_amd.Context(W, H, J)
ctx.set_state(map=gen.random_map(W * H, J, seed=1))
h = ctx._h
fp = X.ctypes.data_as(C.POINTER(C.c_float))
lb = np.zeros(B, np.uint64)
lbp = lb.ctypes.data_as(C.POINTER(C.c_uint64))
mse = C.c_float()
names = ("prefetch_chunk", "commit_chunk", "train_online_chunk_acc", "get_last_bmu", "get_mse")
for sigma0 in (8.0, 1.0, -8.0):          # negative: a new sigma every epoch, as the schedule of Som.cpp:1146 has it
    tot = np.zeros(5)
    for e in range(330):
        sigma = sigma0 if sigma0 > 0 else -sigma0 - 0.001 * e
        t = [time.perf_counter()]
        assert L.vsom_prefetch_chunk(h, fp, C.c_size_t(B)) == 0; t.append(time.perf_counter())
        assert L.vsom_commit_chunk(h) == 0; t.append(time.perf_counter())
        assert L.vsom_train_online_chunk_acc(h, C.c_double(0.01), C.c_double(sigma), 0, 1, None) == 0; t.append(time.perf_counter())
        assert L.vsom_get_last_bmu(h, lbp) == 0; t.append(time.perf_counter())
        assert L.vsom_get_mse(h, C.byref(mse)) == 0; t.append(time.perf_counter())
        if e >= 30:
            tot += np.diff(t)
    tot *= 1e6 / 300
    print(f"sigma {sigma0}: " + ", ".join(f"{n} {v:.1f}" for n, v in zip(names, tot)) + f"  | sum {tot.sum():.1f} us per epoch", flush=True)
ctx.close()
