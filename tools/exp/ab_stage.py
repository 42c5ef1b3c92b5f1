#!/usr/bin/env python3
"""A/B inside ONE process, alternating: C3 batch steps with every chunk staged at the start of its step (A) against chunk
i+1 staged beside the chains of chunk i (B: vsom_stage_next_device + vsom_commit_chunk).  Box-to-box and run-to-run
differences are +-2 %, so the two are interleaved several times and the per-mode medians compared.
  python tools/exp/ab_stage.py [--map 128] [--steps 30] [--rounds 5] [--timers none|update|all]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gen, vsom_amd
from vsom_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--map", type=int, default=128)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--timers", default="update")
a = ap.parse_args()
torch.cuda.set_device(0)
W = a.map; J = 784; B = 4096; sigma = W / 4.0
ctx = vsom_amd.Context(W, W, J, 0)
ctx.set_state(map=gen.random_map(W * W, J, 42) * np.float32(100) + np.float32(100))
chunks = [torch.from_numpy(gen.mnist_like(B, seed=3 + i, dim=J)).cuda() for i in range(4)]
torch.cuda.synchronize()
groups = {"none": [], "update": ["update"], "all": capi.TIMER_NAMES}[a.timers]

def run(mode, n):
    if mode == "B":
        ctx.stage_next_device(chunks[0].data_ptr(), B)
    for i in range(n):
        if mode == "B":
            ctx.commit_chunk()
            ctx.batch_epoch_async(sigma, True)
            ctx.stage_next_device(chunks[(i + 1) % 4].data_ptr(), B)
        else:
            ctx.set_chunk_device(chunks[i % 4].data_ptr(), B)
            ctx.batch_epoch_async(sigma, True)
    ctx.synchronize()

res = {"A": [], "B": []}
upd = {"A": [], "B": []}
for m in ("A", "B"):
    run(m, 5)
for r in range(a.rounds):
    for m in ("A", "B") if r % 2 == 0 else ("B", "A"):
        run(m, 3)
        ctx.get_timing(reset=True)
        ctx.enable_timing(True, groups=groups)
        t0 = time.perf_counter()
        run(m, a.steps)
        dt = time.perf_counter() - t0
        tm = ctx.get_timing(reset=True)
        ctx.enable_timing(False)
        res[m].append(dt / a.steps * 1e3)
        upd[m].append(tm["update"][0] / max(tm["update"][1], 1))
print(json.dumps({"map": W, "timers": a.timers, "A_stage_in_step_ms": [round(x, 4) for x in res["A"]],
                  "B_stage_ahead_ms": [round(x, 4) for x in res["B"]],
                  "A_median": round(float(np.median(res["A"])), 4), "B_median": round(float(np.median(res["B"])), 4),
                  "A_update_ms": round(float(np.median(upd["A"])), 4), "B_update_ms": round(float(np.median(upd["B"])), 4)}))
ctx.close()
