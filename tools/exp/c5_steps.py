#!/usr/bin/env python3
"""C5 (32x32 CLR, J = 64, B = 8192): per step the search's time, the shortlist's redo / candidate counts and mode"""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd
from vsom_amd import capi
W, J, B, sigma = 32, 64, 8192, 8.0
ctx = vsom_amd.Context(W, W, J, capi.CLR)
ctx.set_state(map=gen.random_map(W * W, J * (J - 1), 42))
chunks = [gen.correlated(B, J, seed=3 + i) for i in range(4)]
mode = {"auto": capi.BMU_AUTO, "exact": capi.BMU_EXACT, "shortlist": capi.BMU_SHORTLIST}[sys.argv[1] if len(sys.argv) > 1 else "auto"]
ctx.set_bmu_mode(mode)
for i in range(int(os.environ.get("STEPS", "24"))):
    ctx.upload_chunk(chunks[i % 4])
    ctx.get_timing(reset=True)
    ctx.enable_timing(True, groups=["bmu", "update"])
    ctx.batch_epoch_async(sigma, True)
    ctx.synchronize()
    tm = ctx.get_timing(reset=True)
    st = ctx.shortlist_stats()
    m = ctx.get_state()["map"]
    P = m.shape[1] // 2
    A2 = (m[:, :P].astype(np.float64) ** 2).max(axis=1)
    nB = (m[:, P:].astype(np.float64) ** 2).sum(axis=1)
    extra = {"A2max_pct": [float(np.percentile(A2, q)) for q in (50, 90, 99, 100)],
             "nB_pct": [float(np.percentile(nB, q)) for q in (50, 90, 99, 100)]}
    print(json.dumps({"step": i, **extra, "bmu_ms": round(tm["bmu"][0], 3), "update_ms": round(tm["update"][0], 3), "stats": st,
                      "nan_rows": int(np.isnan(m).any(axis=1).sum()), "distinct_rows": int(len(np.unique(m.round(4), axis=0))),
                      "bit_distinct_rows": int(len(np.unique(np.ascontiguousarray(m).view(np.uint32), axis=0)))}))
ctx.close()
