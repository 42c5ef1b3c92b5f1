#!/usr/bin/env python3
"""Development timing of ONE rank's share of an N-GPU weak-scaling step on a single GPU:
chunk of 4096*N samples resident, phase 1 on this rank's 4096 samples, phase 2 on its N/world
nodes (the collectives are not simulated).  Shows how the per-rank kernels behave when the node
shard shrinks and the chains get longer."""
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

W, D, sigma = 128, 784, 32.0
init = gen.random_map(W * W, D, 42) * np.float32(100) + np.float32(100)
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    B = 4096 * world
    X = gen.mnist_like(B, 3, D)
    ctx = vsom_amd.Context(W, W, D)
    ctx.set_state(map=init)
    if os.environ.get("VSOM_SIM_FMA"):          # the contracted update arithmetic (bench.py's default)
        ctx.set_update_mode(1)
    ctx.upload_chunk(X)
    n1 = W * W // world
    steps = int(os.environ.get("VSOM_SIM_STEPS", "5"))

    def step():
        ctx.batch_phase1_async(0, 4096, True)
        ctx.batch_finish_async()
        ctx.batch_phase2_async(sigma, 0, n1)

    step()
    ctx.synchronize()
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tm = ctx.get_timing(reset=True)
    print(json.dumps({"world": world, "B_total": B, "nodes_rank": n1, "ms_per_step": round(dt * 1e3, 3),
                      "kernel_ms": {k: round(v[0] / steps, 4) for k, v in tm.items() if v[1]}}), flush=True)
    ctx.close()
