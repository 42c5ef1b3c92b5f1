#!/usr/bin/env python3
"""time of one online chunk (20 rows) on the 10 x 10 x 9 fixture shape: the one-launch chunk (VSOM_BMU_AUTO) against the
per-sample kernels (VSOM_BMU_EXACT), HIP events around the chunk's kernels and wall time of the synchronous call"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import gen, vsom_amd
from vsom_amd import capi
for (W, H, J, B) in ((10, 10, 9, 20), (10, 10, 9, 200), (32, 32, 4, 200), (16, 16, 16, 200)):
    X = gen.blobs(B, J, 4, 1, 2, sigma=0.3)
    for mode, name in ((capi.BMU_AUTO, "one launch"), (capi.BMU_EXACT, "per sample")):
        ctx = vsom_amd.Context(W, H, J)
        ctx.set_bmu_mode(mode)
        ctx.set_state(map=gen.random_map(W * H, J, seed=1))
        ctx.upload_chunk(X)
        ctx.train_online_chunk(0.01, 8.0, 0)
        best_dev, best_wall = 1e9, 1e9
        for rep in range(5):
            ctx.get_timing(reset=True)
            ctx.enable_timing(True, groups=("online",))
            t0 = time.perf_counter()
            ctx.train_online_chunk(0.01, 8.0, 0)
            wall = (time.perf_counter() - t0) * 1e6
            tm = ctx.get_timing(reset=True)
            ctx.enable_timing(False)
            best_dev = min(best_dev, tm["online"][0] * 1e3)
            best_wall = min(best_wall, wall)
        print(f"{W}x{H}x{J} B={B} {name}: device {best_dev:.1f} us ({best_dev / B:.2f} per sample), wall {best_wall:.1f} us", flush=True)
        ctx.close()
