#!/usr/bin/env python3
"""Development: does the unequal zero-quad share of a workgroup's 8 wavefronts (barrier skew) cost the chain kernel time?
Times phase 2 at C3 on (a) MNIST-like rows and (b) rows with the SAME live columns and the same overall share of all-zero
(sample, quad) blocks, but spread uniformly over the quads (no systematic difference between wavefronts)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd

W, D, B, sigma = 128, 784, 4096, 32.0
X = gen.mnist_like(B, 3, D)
live = np.flatnonzero((X != 0).any(axis=0))
nq = (live.size + 3) // 4
P = np.zeros((B, nq * 4), np.float32); P[:, :live.size] = X[:, live]
zfrac = float((P.reshape(B, nq, 4) == 0).all(axis=2).mean())
rs = np.random.RandomState(1)
U = np.zeros((B, nq * 4), np.float32)
nzq = rs.rand(B, nq) >= zfrac
vals = (rs.randint(1, 256, size=(B, nq, 4))).astype(np.float32)
U = (vals * nzq[:, :, None]).reshape(B, nq * 4)
Xu = np.zeros_like(X); Xu[:, live] = U[:, :live.size]
init = gen.random_map(W * W, D, 42) * np.float32(100) + np.float32(100)
for name, Xc in (("mnist_like", X), ("uniform_zero_quads", Xu), ("mnist_like", X), ("uniform_zero_quads", Xu)):
    ctx = vsom_amd.Context(W, W, D)
    ctx.set_state(map=init); ctx.upload_chunk(Xc)
    ctx.batch_epoch(sigma, True); ctx.synchronize()
    ctx.enable_timing(True); ctx.get_timing(reset=True)
    for _ in range(10):
        ctx.batch_phase2_async(sigma, 0, W * W)
    ctx.synchronize()
    tm = ctx.get_timing(reset=True)
    lq = (Xc[:, live] != 0)
    print(json.dumps({"data": name, "zero_quad_frac": round(float(((np.pad(Xc[:, live], ((0,0),(0,nq*4-live.size))).reshape(B,nq,4))==0).all(axis=2).mean()),4),
                      "update_ms": round(tm["update"][0] / 10, 4)}), flush=True)
    ctx.close()
