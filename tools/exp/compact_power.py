#!/usr/bin/env python3
"""Does retiring dead columns shorten the phase-2 launch?  C3 shape, strict arithmetic, two data sets:
`window` (rounds 1-2 generator: 400 of 784 columns live) and `strokes` (tests/gen.mnist_like: 661 live).
Run once as is and once with VSOM_NO_COMPACT=1; under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU` the
cycle count of the update kernel gives the clock it ran at."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import vsom_amd  # noqa: E402

W, D, B, sigma = 128, 784, 4096, 32.0
init = gen.random_map(W * W, D, 42) * np.float32(100) + np.float32(100)
steps = int(os.environ.get("VSOM_EXP_STEPS", "10"))
for name in sys.argv[1:] or ["window", "strokes"]:
    X = gen.mnist_like_window(B, 3, D) if name == "window" else gen.mnist_like(B, 3, D)
    for mode in (0, 2, 1):
        ctx = vsom_amd.Context(W, W, D)
        ctx.set_state(map=init)
        ctx.set_update_mode(mode)
        ctx.upload_chunk(X)
        ctx.batch_epoch(sigma, True)
        ctx.enable_timing(True)
        ctx.get_timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.upload_chunk(X)
            ctx.batch_epoch_async(sigma, True)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        tm = ctx.get_timing(reset=True)
        print(json.dumps({"data": name, "live": gen.column_occupancy(X)[0], "mode": mode,
                          "compact": os.environ.get("VSOM_NO_COMPACT", "0") != "1", "ms_per_step": round(dt * 1e3, 3),
                          "kernel_ms": {k: round(v[0] / steps, 4) for k, v in tm.items() if v[1]}}), flush=True)
        ctx.close()
