#!/usr/bin/env python3
"""Whole trainBatchSom schedule (Som.cpp:716-754) at BASELINE config 3's size on one GPU: sigma = sigma0 exp(-decay e)
down to 1, two chunks of 4096 MNIST-like rows per epoch, one line per epoch (sigma, ms per chunk, share of NaN in the
map).  Late epochs are where the neighbourhood table underflows to exact zeros and most of the map turns NaN (a node no
sample of the chunk reaches has W = 0) -- the measurement behind profiles/r4_late_epoch_skip_experiment.txt."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gen            # noqa: E402
import vsom_amd       # noqa: E402
from vsom_amd import capi   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", type=int, default=128)
    ap.add_argument("--dim", type=int, default=784)
    ap.add_argument("--rows", type=int, default=8192)
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--sigma0", type=float, default=32.0)
    ap.add_argument("--decay", type=float, default=0.1)
    ap.add_argument("--epochs", type=int, default=40)
    a = ap.parse_args()
    W = a.map
    X = gen.mnist_like(a.rows, 3, a.dim)
    init = gen.random_map(W * W, a.dim, 42) * np.float32(100)
    ctx = vsom_amd.Context(W, W, a.dim, capi.STANDARD)
    ctx.set_state(map=init)
    total = 0.0
    for e in range(a.epochs):
        sigma = a.sigma0 * math.exp(-a.decay * e)
        if sigma < 1.0:
            break
        ctx.synchronize()
        t0 = time.perf_counter()
        n = 0
        for c0 in range(0, a.rows, a.chunk):
            ctx.upload_chunk(X[c0:c0 + a.chunk])
            ctx.batch_epoch_async(sigma, e == 0)
            n += 1
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        total += dt * n
        m = ctx.get_state()["map"]
        print(json.dumps({"epoch": e, "sigma": round(sigma, 3), "ms_per_chunk_incl_upload": round(dt, 3),
                          "nan_share_of_map": round(float(np.isnan(m).mean()), 4)}), flush=True)
    print(json.dumps({"schedule_ms": round(total, 1)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
