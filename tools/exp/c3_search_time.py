#!/usr/bin/env python3
"""kernel-group time of the C3 search alone (full search of one resident chunk, repeated)"""
import os, sys, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd
W, J, B = 128, 784, 4096
ctx = vsom_amd.Context(W, W, J, 0)
X = gen.mnist_like(B, 3, J)
init = gen.random_map(W * W, J, 42) * np.float32(100) + np.float32(100)
ctx.set_state(map=init)
ctx.upload_chunk(X)
ctx.batch_epoch(32.0, True)          # a trained-looking map
ctx.upload_chunk(X)
for _ in range(3):
    ctx.batch_phase1_async(0, B, True)
ctx.synchronize()
ctx.get_timing(reset=True)
ctx.enable_timing(True, groups=["bmu"])
for _ in range(20):
    ctx.batch_phase1_async(0, B, True)
ctx.synchronize()
tm = ctx.get_timing(reset=True)
print(json.dumps({"bmu_ms": round(tm["bmu"][0] / tm["bmu"][1], 4)}))
