#!/usr/bin/env python3
"""Development timing of every BASELINE.json configuration (per-kernel-group HIP-event times).

Not the contract benchmark (that is bench.py, C3); this one exists to find the weak kernels of
the other configurations: C2 (64x64x784 standard), C3 later epochs (findLocalBmu), C4 (median),
C5 (CLR), and the online path at C3's size.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import vsom_amd  # noqa: E402
from vsom_amd import capi  # noqa: E402


def batch_case(name, W, J, tr, B, sigma, X, init, steps=10, is_first=True, flops_upd=6.0, bmu_mode=None):
    ctx = vsom_amd.Context(W, W, J, tr)
    if bmu_mode is not None:
        ctx.set_bmu_mode(bmu_mode)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    for _ in range(2):
        ctx.batch_epoch(sigma, is_first)
        ctx.set_state(map=init)
        ctx.upload_chunk(X)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.batch_epoch_async(sigma, is_first)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = ctx.get_timing(reset=True)
    D = ctx.depth
    N = W * W
    out = {"case": name, "N": N, "D": D, "B": B, "ms_per_step": round(dt * 1e3, 4),
           "samples_per_s": round(B / dt, 1),
           "kernel_ms": {k: round(v[0] / steps, 4) for k, v in tm.items() if v[1]}}
    upd = tm["update"][0] / steps / 1e3
    if upd > 0:
        out["update_tflops"] = round(flops_upd * N * D * B / upd / 1e12, 2)
    print(json.dumps(out), flush=True)
    ctx.close()


def online_case(name, W, J, tr, B, sigma, X, init, decay):
    ctx = vsom_amd.Context(W, W, J, tr)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    ctx.train_online_chunk(0.1, sigma, decay)
    ctx.upload_chunk(X)
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    ctx.train_online_chunk(0.1, sigma, decay)
    dt = time.perf_counter() - t0
    tm = ctx.get_timing(reset=True)
    N, D = W * W, ctx.depth
    k = min(W, int(2.5 * sigma) * 2 + 1) ** 2
    bytes_per_sample = 4.0 * N * D + 20.0 * k * D
    print(json.dumps({"case": name, "N": N, "D": D, "B": B, "sigma": sigma,
                      "us_per_sample": round(dt / B * 1e6, 2),
                      "kernel_us_per_sample": round(tm["online"][0] / B * 1e3, 2),
                      "approx_window_nodes": k,
                      "approx_GBps": round(bytes_per_sample / (dt / B) / 1e9, 1)}), flush=True)
    ctx.close()


def main():
    which = set(sys.argv[1:]) or {"c1", "c2", "c2median", "c3local", "c4", "c5", "online"}
    if "c1" in which:
        X = gen.blobs(1024, 16, 4, 1, 2, sigma=0.1)
        init = gen.random_map(100, 16, 42)
        batch_case("C1 10x10x16 std first", 10, 16, capi.STANDARD, 1024, 5.0, X, init, steps=50)
        batch_case("C1 10x10x16 std local", 10, 16, capi.STANDARD, 1024, 5.0, X, init, steps=50, is_first=False)
    if "c2" in which:
        X = gen.mnist_like(4096, 3, 784)
        init = gen.random_map(64 * 64, 784, 42) * np.float32(100) + np.float32(100)
        batch_case("C2 64x64x784 std first", 64, 784, capi.STANDARD, 4096, 16.0, X, init)
    if "c2median" in which:
        X = gen.mnist_like(4096, 3, 784)
        init = gen.random_map(64 * 64, 784, 42) * np.float32(100) + np.float32(100)
        batch_case("C2m 64x64x784 median first", 64, 784, capi.MEDIAN, 4096, 16.0, X, init)
        init = gen.random_map(128 * 128, 784, 42) * np.float32(100) + np.float32(100)
        batch_case("C3m 128x128x784 median first", 128, 784, capi.MEDIAN, 4096, 32.0, X, init, steps=5)
    if "c3exact" in which:
        X = gen.mnist_like(4096, 3, 784)
        init = gen.random_map(128 * 128, 784, 42) * np.float32(100) + np.float32(100)
        batch_case("C3 128x128x784 std, exact-order search kernel", 128, 784, capi.STANDARD, 4096, 32.0, X, init, steps=5,
                   bmu_mode=capi.BMU_EXACT)
    if "c3local" in which:
        X = gen.mnist_like(4096, 3, 784)
        init = gen.random_map(128 * 128, 784, 42) * np.float32(100) + np.float32(100)
        batch_case("C3 128x128x784 std local", 128, 784, capi.STANDARD, 4096, 32.0, X, init, is_first=False)
    if "c4" in which:
        X = gen.blobs(16384, 32, 8, 1, 4, sigma=1.0)
        init = gen.random_map(64 * 64, 32, 42)
        batch_case("C4 64x64x32 median first", 64, 32, capi.MEDIAN, 16384, 16.0, X, init, flops_upd=6.0)
        batch_case("C4 64x64x32 median local", 64, 32, capi.MEDIAN, 16384, 16.0, X, init, is_first=False)
        batch_case("C4' 64x64x32 std first", 64, 32, capi.STANDARD, 16384, 16.0, X, init)
        batch_case("C4 64x64x32 median first, exact-order search kernel", 64, 32, capi.MEDIAN, 16384, 16.0, X, init,
                   bmu_mode=capi.BMU_EXACT)
        for dd in (64, 128):
            Xd = gen.blobs(16384, dd, 8, 1, 4, sigma=1.0)
            initd = gen.random_map(64 * 64, dd, 42)
            batch_case(f"64x64x{dd} std first (auto)", 64, dd, capi.STANDARD, 16384, 16.0, Xd, initd)
            batch_case(f"64x64x{dd} std first (exact)", 64, dd, capi.STANDARD, 16384, 16.0, Xd, initd, bmu_mode=capi.BMU_EXACT)
    if "c5" in which:
        X = gen.correlated(8192, 64, 5)
        D = 64 * 63
        init = gen.random_map(32 * 32, D, 42)
        # update ~16 flop per (node, pair, sample) = 8 per model element
        batch_case("C5 32x32 J=64 CLR first", 32, 64, capi.CLR, 8192, 8.0, X, init, steps=5, flops_upd=8.0)
    if "online" in which:
        X = gen.mnist_like(512, 3, 784)
        init = gen.random_map(128 * 128, 784, 42) * np.float32(100) + np.float32(100)
        for sigma in (32.0, 8.0, 2.0, 1.0):
            online_case("online C3 exp", 128, 784, capi.STANDARD, 512, sigma, X, init, capi.EXPONENTIAL)
        online_case("online C3 inv", 128, 784, capi.STANDARD, 512, 8.0, X, init, capi.INVERSE_PROPORTIONAL)
        X = gen.blobs(2048, 32, 8, 1, 4, sigma=1.0)
        init = gen.random_map(64 * 64, 32, 42)
        online_case("online C4 median", 64, 32, capi.MEDIAN, 2048, 4.0, X, init, capi.EXPONENTIAL)


if __name__ == "__main__":
    main()
