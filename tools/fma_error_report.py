#!/usr/bin/env python3
"""How far the opt-in contracted update arithmetic (VSOM_UPDATE_FMA) is from the oracle at C3."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po
W, D, B, sigma = 128, 784, 4096, 32.0
X = gen.mnist_like(B, 3, D)
init = gen.random_map(W * W, D, 42) * np.float32(100) + np.float32(100)
o = po.OracleSom(W, W, D); o.set_state(map=init)
lb = np.zeros(B, np.uint64); o.batch_epoch(X, lb, sigma, True, nthreads=min(128, po.max_threads()))
ctx = vsom_amd.Context(W, W, D); ctx.set_state(map=init); ctx.set_update_mode(capi.UPDATE_FMA)
ctx.upload_chunk(X); ctx.batch_epoch(sigma, True); st = ctx.get_state()
for k, ref in (("map", o.map), ("sigma", o.sigma)):
    a, b = st[k].astype(np.float64), ref.astype(np.float64)
    rowmax = np.maximum(np.abs(b).max(axis=1, keepdims=True), 1e-30)
    nz = np.abs(b) > 0
    print(k, "max |err| / row max:", float((np.abs(a - b) / rowmax).max()),
          " max elementwise relative (nonzero ref):", float((np.abs(a - b)[nz] / np.abs(b)[nz]).max()),
          " differing elements:", int((st[k].view(np.uint32) != ref.view(np.uint32)).sum()), "of", b.size)
print("BMU identical:", bool((ctx.get_last_bmu() == lb).all()), " weightMap identical:", bool((st["weight"].view(np.uint32) == o.weight.view(np.uint32)).all()))
