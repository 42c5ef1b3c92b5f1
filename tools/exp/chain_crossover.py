#!/usr/bin/env python3
"""Development: update-kernel time of lane = node assembly vs the LDS-staged chain kernel over map shapes
(run once with VSOM_CHAIN_MAX_WAVES=0 and once with a huge value)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd
from vsom_amd import capi
fma = os.environ.get("VSOM_SIM_FMA") == "1"
for (W, H, D, B, tr) in [(64, 64, 128, 8192, 0), (64, 64, 256, 8192, 0), (64, 64, 384, 8192, 0), (64, 64, 512, 8192, 0),
                         (64, 64, 784, 4096, 0), (32, 32, 784, 8192, 0), (32, 32, 2048, 4096, 0), (128, 128, 64, 8192, 0),
                         (128, 128, 128, 4096, 0), (64, 64, 256, 8192, 1), (96, 96, 200, 4096, 1)]:
    X = gen.blobs(B, D, 8, 1, 4, sigma=1.0)
    ctx = vsom_amd.Context(W, H, D, tr)
    ctx.set_state(map=gen.random_map(W * H, D, 42))
    if fma:
        ctx.set_update_mode(1)
    ctx.upload_chunk(X)
    ctx.batch_epoch(max(W, H) / 4.0, True)
    ctx.enable_timing(True); ctx.get_timing(reset=True)
    for _ in range(4):
        ctx.batch_epoch_async(max(W, H) / 4.0, True)
    ctx.synchronize()
    tm = ctx.get_timing(reset=True)
    waves = ((W * H + 63) // 64) * ((D + 13) // 14)
    print(json.dumps({"N": W * H, "D": D, "B": B, "tr": tr, "lane_node_waves": waves, "update_ms": round(tm["update"][0] / 4, 4),
                      "sigma_ms": round(tm["sigma"][0] / 4, 4)}), flush=True)
    ctx.close()
