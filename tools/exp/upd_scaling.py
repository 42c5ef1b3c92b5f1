#!/usr/bin/env python3
"""Development: update-kernel time against (node shard, chunk length) on one GPU."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd  # noqa: E402
W, D, sigma = 128, int(os.environ.get("VSOM_D", "784")), 32.0
init = gen.random_map(W * W, D, 42) * np.float32(100) + np.float32(100)
for spec in sys.argv[1:]:
    nloc, B = (int(v) for v in spec.split(","))
    X = gen.mnist_like(B, 3, D)
    ctx = vsom_amd.Context(W, W, D)
    ctx.set_state(map=init)
    if os.environ.get("VSOM_SIM_FMA"):
        ctx.set_update_mode(1)
    ctx.upload_chunk(X)
    ctx.batch_phase1_async(0, min(B, 4096), True)
    ctx.batch_finish_async()
    for _ in range(2):
        ctx.batch_phase2_async(sigma, 0, nloc)
    ctx.synchronize()
    ctx.enable_timing(True); ctx.get_timing(reset=True)
    steps = 10
    for _ in range(steps):
        ctx.batch_phase2_async(sigma, 0, nloc)
    ctx.synchronize()
    tm = ctx.get_timing(reset=True)
    upd = tm["update"][0] / steps
    print(json.dumps({"nloc": nloc, "B": B, "update_ms": round(upd, 4), "cw_ms": round(tm["cw"][0] / steps, 4),
                      "ns_per_node_sample_dim": round(upd * 1e6 / (nloc * B * D), 6),
                      "tflops": round(6.0 * nloc * B * D / upd / 1e9, 2)}), flush=True)
    ctx.close()
