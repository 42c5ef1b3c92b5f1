#!/usr/bin/env python3
"""Does a non-strict update arithmetic stay on the reference's trajectory over a whole trainBatchSom
schedule (Som.cpp:716-754: first epoch findBmu, later epochs findLocalBmu on the map the previous chunk
wrote, :786-805)?

Runs the schedule chunk by chunk on the device in the given arithmetic and in the strict oracle, each on
ITS OWN map, and prints per epoch: samples whose lastBMU differs, worst element-wise relative error of map /
sigmaMap, elements over 1e-5.  The first line whose `bmu_diff` is non-zero is where the run has left the
reference's trajectory.

usage: fma_schedule_report.py [--mode fma|fma_sigma|strict] [case ...]      cases: s24 c2 c3"""
import argparse
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import vsom_amd  # noqa: E402
from vsom_amd import capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

#        W    J    rows  chunk  sigma0 decay epochs
CASES = {
    "s24": (24, 784, 2048, 1024, 8.0, 0.1, 14),
    "c2": (64, 784, 8192, 4096, 16.0, 0.1, 12),
    "c3": (128, 784, 8192, 4096, 32.0, 0.1, 10),
}


def elementwise(a, b):
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    ok = np.isfinite(b64) & (b64 != 0)
    rel = np.abs(a64 - b64)[ok] / np.abs(b64[ok])
    nan_same = bool((np.isnan(a64) == np.isnan(b64)).all())
    return (float(rel.max()) if rel.size else 0.0), int((rel > 1e-5).sum()), nan_same


def run(name, mode):
    W, J, rows, chunk, sigma0, decay, epochs = CASES[name]
    X = gen.mnist_like(rows, 3, J)
    init = gen.random_map(W * W, J, 42) * np.float32(100)
    thr = max(1, min(128, po.max_threads()))
    o = po.OracleSom(W, W, J, po.STANDARD)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, W, J, capi.STANDARD)
    ctx.set_state(map=init)
    ctx.set_update_mode(mode)
    first_div = None
    for e in range(epochs):
        sigma = sigma0 * math.exp(-decay * e)
        if sigma < 1.0:
            break
        rec = {"case": name, "mode": int(mode), "epoch": e, "sigma": round(sigma, 4), "bmu_diff": 0, "samples": 0,
               "map_max_rel": 0.0, "map_over_1e-5": 0, "sigma_max_rel": 0.0, "sigma_over_1e-5": 0,
               "mse_equal": True, "weight_equal": True}
        for c0 in range(0, rows, chunk):
            Xc = X[c0:c0 + chunk]
            lb = np.zeros(Xc.shape[0], np.uint64)
            mse_o = o.batch_epoch(Xc, lb, sigma, e == 0, nthreads=thr)
            ctx.upload_chunk(Xc)
            mse_g = ctx.batch_epoch(sigma, e == 0)
            st = ctx.get_state(S=False)
            rec["bmu_diff"] += int((ctx.get_last_bmu() != lb).sum())
            rec["samples"] += int(lb.size)
            rec["mse_equal"] &= bool(np.float32(mse_g) == np.float32(mse_o))
            rec["weight_equal"] &= bool((st["weight"].view(np.uint32) == o.weight.view(np.uint32)).all())
            for k, ref in (("map", o.map), ("sigma", o.sigma)):
                worst, over, _ = elementwise(st[k], ref)
                rec[k + "_max_rel"] = max(rec[k + "_max_rel"], worst)
                rec[k + "_over_1e-5"] += over
        if rec["bmu_diff"] and first_div is None:
            first_div = e
        print(json.dumps(rec), flush=True)
    print(json.dumps({"case": name, "mode": int(mode), "first_epoch_with_bmu_difference": first_div}), flush=True)
    ctx.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="fma")
    ap.add_argument("cases", nargs="*")
    a = ap.parse_args()
    modes = {"strict": capi.UPDATE_STRICT, "fma": capi.UPDATE_FMA}
    if hasattr(capi, "UPDATE_FMA_SIGMA"):
        modes["fma_sigma"] = capi.UPDATE_FMA_SIGMA
    for n in (a.cases or ["s24", "c2"]):
        run(n, modes[a.mode])
