#!/usr/bin/env python3
"""Development timing of the online path (trainBasicSom inner loop) at BASELINE's map size."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
D = int(sys.argv[2]) if len(sys.argv) > 2 else 784
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
for sigma in (32.0, 8.0, 2.0, 1.0):
    ctx = vsom_amd.Context(W, H, D)
    ctx.set_state(map=gen.random_map(W * H, D, 42) * np.float32(100) + np.float32(100))
    X = gen.mnist_like(B, 3, D)
    ctx.upload_chunk(X)
    ctx.train_online_chunk(0.1, sigma, 0)
    ctx.upload_chunk(X)
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    ctx.train_online_chunk(0.1, sigma, 0)
    dt = time.perf_counter() - t0
    tm = ctx.get_timing()
    print(f"sigma={sigma}: {B/dt:.0f} samples/s  ({dt/B*1e6:.1f} us/sample wall, online kernels {tm['online'][0]/B*1e3:.1f} us/sample)")
    ctx.close()
