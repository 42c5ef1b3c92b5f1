#!/usr/bin/env python3
"""Online chunk loop (trainBasicSom's inner loop), exact scan vs image-bounded search, over map sizes and sigmas:
one JSON line per (map, depth, sigma) with the microseconds per sample of both (HIP events around the chunk's kernels)
and what VSOM_BMU_AUTO picks.  python tools/online_sweep.py [B]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: one HIP runtime per process, tests/conftest.py)
import gen  # noqa: E402
import vsom_amd  # noqa: E402
from vsom_amd import capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
CASES = [(128, 784), (96, 784), (64, 784), (48, 784), (32, 784), (128, 256), (128, 64), (256, 128), (192, 784)]
SIGMAS = [32.0, 16.0, 8.0, 4.0, 2.0]


def run(W, D, sigma, mode):
    ctx = vsom_amd.Context(W, W, D)
    ctx.set_bmu_mode(mode)
    ctx.set_state(map=gen.random_map(W * W, D, 42) * np.float32(100) + np.float32(100))
    X = gen.mnist_like(B, 3, D)
    ctx.upload_chunk(X)
    ctx.train_online_chunk(0.1, sigma, 0)            # warm-up (allocations, tables)
    best = None
    for rep in range(3):
        ctx.upload_chunk(X)
        ctx.get_timing(reset=True)
        ctx.enable_timing(True, groups=("online",))
        ctx.train_online_chunk(0.1, sigma, 0)
        tm = ctx.get_timing(reset=True)
        ctx.enable_timing(False)
        us = tm["online"][0] / B * 1e3
        best = us if best is None else min(best, us)
    st = ctx.online_search_stats(reset=True)
    ctx.close()
    return best, st


for W, D in CASES:
    for sigma in SIGMAS:
        if 2.5 * sigma > 2 * W:
            continue
        ex, _ = run(W, D, sigma, capi.BMU_EXACT)
        im, st = run(W, D, sigma, capi.BMU_SHORTLIST)
        au, sta = run(W, D, sigma, capi.BMU_AUTO)
        print(json.dumps({"map": W, "depth": D, "sigma": sigma, "chunk": B, "exact_us_per_sample": round(ex, 2),
                          "image_us_per_sample": round(im, 2), "image_over_exact": round(im / ex, 3),
                          "auto_us_per_sample": round(au, 2), "auto_picks": "image" if sta["samples"] else "exact",
                          "exact_evaluations_per_sample": round(st["exact_evaluations"] / max(st["samples"], 1), 1)}), flush=True)
