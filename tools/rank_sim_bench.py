#!/usr/bin/env python3
"""Development timing of ONE rank's share of an N-GPU step on a single GPU (the collectives are not
simulated): shows how the per-rank kernels behave when the node shard shrinks.

  weak   (VSOM_SIM_SPLIT=weak):   chunk of 4096*N samples resident, phase 1 on this rank's 4096, phase 2 on its
                                  16384/N nodes over all 4096*N samples;
  strong (default, BASELINE config 3: the 4096-sample chunk sharded across the GPUs): chunk of 4096 resident,
                                  phase 1 on this rank's 4096/N samples, phase 2 on its 16384/N nodes.
VSOM_SIM_ARITH=strict|sigma|contracted (default strict).  usage: rank_sim_bench.py [N ...]"""
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
split = os.environ.get("VSOM_SIM_SPLIT", "strong")
arith = os.environ.get("VSOM_SIM_ARITH", "contracted" if os.environ.get("VSOM_SIM_FMA") else "strict")
init = gen.random_map(W * W, D, 42) * np.float32(100) + np.float32(100)
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    B = 4096 * world if split == "weak" else 4096
    s1 = 4096 if split == "weak" else 4096 // world
    X = gen.mnist_like(B, 3, D)
    ctx = vsom_amd.Context(W, W, D)
    ctx.set_state(map=init)
    ctx.set_update_mode({"strict": 0, "contracted": 1, "sigma": 2}[arith])
    ctx.upload_chunk(X)
    n1 = W * W // world
    steps = int(os.environ.get("VSOM_SIM_STEPS", "10"))

    def step():
        ctx.batch_phase1_async(0, s1, True)
        ctx.batch_finish_async()
        ctx.batch_phase2_async(sigma, 0, n1)

    # `ms_per_step` from a pass that times the chain kernel's group only (every timed group costs two event records between
    # kernels that otherwise run back to back: ~10 us of idle device each; all seven were 8 % of a rank's 0.9 ms step at
    # N = 8), the per-group breakdown from a second pass with all of them
    step()
    ctx.synchronize()
    ctx.enable_timing(True, groups=["update"])
    ctx.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    for _ in range(steps):
        step()
    ctx.synchronize()
    tm = ctx.get_timing(reset=True)
    print(json.dumps({"split": split, "arith": arith, "world": world, "B_total": B, "samples_rank": s1, "nodes_rank": n1,
                      "ms_per_step": round(dt * 1e3, 3),
                      "ideal_ms": None if world == 1 else "ms_per_step(world=1)" + ("" if split == "weak" else f"/{world}"),
                      "kernel_ms": {k: round(v[0] / steps, 4) for k, v in tm.items() if v[1]}}), flush=True)
    ctx.close()
