#!/usr/bin/env python3
"""A/B of one environment switch the library reads at vsom_create, inside ONE process: two contexts created under the two
values, C3-size batch steps timed alternately; per-value medians of the step and of one kernel group.
  python tools/exp/ab_env.py VSOM_SL_RING old two [--group bmu] [--map 128] [--data u8|float]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gen, vsom_amd

ap = argparse.ArgumentParser()
ap.add_argument("var"); ap.add_argument("a"); ap.add_argument("b")
ap.add_argument("--group", default="bmu"); ap.add_argument("--map", type=int, default=128)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--data", default="u8")
A = ap.parse_args()
torch.cuda.set_device(0)
W, J, B = A.map, 784, 4096
sigma = W / 4.0
rows = [gen.mnist_like(B, seed=3 + i, dim=J) for i in range(2)]
init = gen.random_map(W * W, J, 42) * np.float32(100) + np.float32(100)
if A.data == "float":
    rows = [(r / np.float32(255)).astype(np.float32) for r in rows]
    init = (init / np.float32(255)).astype(np.float32)
chunks = [torch.from_numpy(r).cuda() for r in rows]
ctxs = []
for val in (A.a, A.b):
    os.environ[A.var] = val
    c = vsom_amd.Context(W, W, J, 0)
    c.set_state(map=init)
    for i in range(3):
        c.set_chunk_device(chunks[i % 2].data_ptr(), B)
        c.batch_epoch_async(sigma, True)
    c.synchronize()
    ctxs.append(c)
grp, stp = [[], []], [[], []]
for r in range(A.rounds):
    for k in ((0, 1) if r % 2 == 0 else (1, 0)):
        c = ctxs[k]
        c.get_timing(reset=True)
        c.enable_timing(True, groups=[A.group])
        t0 = time.perf_counter()
        for i in range(A.steps):
            c.set_chunk_device(chunks[i % 2].data_ptr(), B)
            c.batch_epoch_async(sigma, True)
        c.synchronize()
        stp[k].append((time.perf_counter() - t0) / A.steps * 1e3)
        tm = c.get_timing(reset=True)
        grp[k].append(tm[A.group][0] / tm[A.group][1])
print(json.dumps({"var": A.var, "map": W, "data": A.data, "group": A.group,
                  A.a: {"step_ms": round(float(np.median(stp[0])), 4), "group_ms": round(float(np.median(grp[0])), 4)},
                  A.b: {"step_ms": round(float(np.median(stp[1])), 4), "group_ms": round(float(np.median(grp[1])), 4)}}))
