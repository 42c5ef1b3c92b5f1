#!/usr/bin/env python3
"""Development: per-epoch shortlist statistics and search time of the CLR search at C5's shape."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen, vsom_amd
from vsom_amd import capi
W, J, B, sigma = 32, 64, 8192, float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
D = J * (J - 1)
chunks = [gen.correlated(B, J, 5 + i) for i in range(4)]
ctx = vsom_amd.Context(W, W, J, capi.CLR)
ctx.set_state(map=gen.random_map(W * W, D, 42))
ctx.enable_timing(True)
for mode, name in ((capi.BMU_SHORTLIST, "shortlist"), (capi.BMU_EXACT, "exact"), (capi.BMU_AUTO, "auto")):
    ctx.set_bmu_mode(mode)
    ctx.set_state(map=gen.random_map(W * W, D, 42))
    for i in range(8):
        ctx.upload_chunk(chunks[i % 4])
        ctx.get_timing(reset=True)
        ctx.batch_epoch(sigma, True)
        tm = ctx.get_timing(reset=True)
        st = ctx.shortlist_stats()
        print(name, i, "bmu_ms %.3f" % tm["bmu"][0], "update_ms %.3f" % tm["update"][0], st, flush=True)
