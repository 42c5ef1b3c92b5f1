#!/usr/bin/env python3
"""How far the contracted update arithmetic (VSOM_UPDATE_FMA) is from the oracle, element by element.

For every case prints, for map and sigmaMap:
  * pure element-wise relative error |a-b|/|b| over the elements with b != 0: max, and how many exceed 1e-5;
  * the error against the scale of the chain's operands, |a-b| / max(|b|, colscale_d) with
    colscale_d = max_j |x_j,d| (what tests/test_gpu_fma_mode.py asserts);
  * how many elements differ at all.
usage: fma_error_report.py [c2] [c3] [blobs] [c4std] ..."""
import json
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

CASES = {
    "c3": (128, 784, 4096, 32.0, "mnist"),
    "c2": (64, 784, 4096, 16.0, "mnist"),
    "blobs": (32, 28, 1500, 10.0, "blobs"),
    "blobs48": (48, 75, 700, 8.0, "blobs"),
    "c4std": (64, 32, 16384, 16.0, "blobs"),
    "clr9": (12, 9, 300, 3.0, "clr"),
    "clr64": (32, 64, 2048, 8.0, "clr"),
}


def report(name):
    W, J, B, sigma, kind = CASES[name]
    tr = capi.CLR if kind == "clr" else capi.STANDARD
    if kind == "clr":
        X = gen.correlated(B, J, 5)
        init = gen.random_map(W * W, capi.model_length(tr, J), 42)
    else:
        X = gen.mnist_like(B, 3, J) if kind == "mnist" else gen.blobs(B, J, 5, 1, 2, sigma=0.4)
        init = gen.random_map(W * W, J, 42) * (np.float32(100) if kind == "mnist" else np.float32(1))
    o = po.OracleSom(W, W, J, tr)
    o.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    o.batch_epoch(X, lb, sigma, True, nthreads=min(128, po.max_threads()))
    ctx = vsom_amd.Context(W, W, J, tr)
    ctx.set_state(map=init)
    ctx.set_update_mode(capi.UPDATE_FMA)
    ctx.upload_chunk(X)
    ctx.batch_epoch(sigma, True)
    st = ctx.get_state()
    col = np.abs(X).max(axis=0, keepdims=True).astype(np.float64)
    if kind == "clr":   # no input column bounds a pair's steps: the node's largest reference magnitude instead
        col = None
    out = {"case": name, "W": W, "J": J, "B": B, "bmu_identical": bool((ctx.get_last_bmu() == lb).all()),
           "weight_identical": bool((st["weight"].view(np.uint32) == o.weight.view(np.uint32)).all())}
    for k, ref, scale in (("map", o.map, col), ("sigma", o.sigma, col)):
        a, b = st[k].astype(np.float64), ref.astype(np.float64)
        if scale is None:
            scale = np.abs(np.where(np.isfinite(b), b, 0.0)).max(axis=1, keepdims=True)
        ok = np.isfinite(b)
        err = np.abs(a - b)
        nz = ok & (np.abs(b) > 0)
        rel = err[nz] / np.abs(b[nz])
        den = np.maximum(np.abs(b), np.broadcast_to(scale, b.shape))
        sc = np.where(den > 0, err / np.where(den > 0, den, 1.0), np.where(err > 0, np.inf, 0.0))
        out[k] = {"max_elementwise_rel": float(rel.max()) if rel.size else 0.0,
                  "elements_over_1e-5": int((rel > 1e-5).sum()), "elements_over_1e-6": int((rel > 1e-6).sum()),
                  "nonzero_elements": int(nz.sum()),
                  "max_err_over_max(|ref|,colscale)": float(sc[ok].max()),
                  "differing": int((st[k].view(np.uint32) != ref.view(np.uint32)).sum()), "elements": int(b.size)}
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["c3", "c2", "blobs", "blobs48", "c4std"]):
        report(n)
