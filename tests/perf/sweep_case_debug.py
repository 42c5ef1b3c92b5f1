#!/usr/bin/env python3
"""Development: re-run one case of tests/test_gpu_random_shapes.py::test_random_shape_assembly_update and show
which samples' BMUs differ between the oracle, the exact search and the shortlist search."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("VSOM_SWEEP_N", "300"); os.environ.setdefault("VSOM_ASM_SWEEP_N", "150"); os.environ.setdefault("VSOM_SWEEP_SEED", "777")
import gen, vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po
import test_gpu_random_shapes as T
want = sys.argv[1]
case = [c for c in T.ASM_CASES if c[0] == want][0]
name, W, H, J, B, sigma, seed = case
tr = po.MEDIAN
rs = np.random.RandomState(seed)
X = (rs.randn(B, J) * rs.choice([0.1, 1.0, 50.0])).astype(np.float32)
X[rs.rand(B, J) < 0.1] = 0.0
init = gen.random_map(W * H, J, seed=seed % 1000)
X[rs.rand(B, J) < 0.02] = -0.0
X[rs.rand(B, J) < 0.01] = np.float32(1e-42)
X[rs.rand(B, J) < 0.01] = np.float32(-3e-45)
X[rs.rand(B, J) < 0.01] = np.float32(3e38)
X[rs.rand(B, J) < 0.002] = np.inf
X[rs.rand(B, J) < 0.002] = -np.inf
X[rs.rand(B, J) < 0.002] = np.nan
init[rs.rand(*init.shape) < 0.05] = 0.0
orc = po.OracleSom(W, H, J, tr); orc.set_state(map=init)
lb = np.zeros(B, np.uint64); sq = np.zeros(B, np.float32)
orc.batch_phase1_range(X, 0, B, lb, sq, True, nthreads=8)
ctx = vsom_amd.Context(W, H, J, tr); ctx.set_state(map=init); ctx.upload_chunk(X)
for mode, nm in ((capi.BMU_EXACT, "exact"), (capi.BMU_SHORTLIST, "shortlist"), (capi.BMU_AUTO, "auto")):
    ctx.set_bmu_mode(mode)
    idx, dist = ctx.bmu_batch()
    bad = np.nonzero(idx != lb)[0]
    print(nm, "mismatching samples:", bad.tolist(), ctx.shortlist_stats())
    for s in bad[:6]:
        x = X[s]
        print("  sample", s, "oracle", lb[s], sq[s], "gpu", idx[s], dist[s], "nan", int(np.isnan(x).sum()), "inf", int(np.isinf(x).sum()),
              "big", int((np.abs(x) > 1e38).sum()), "max finite", float(np.nanmax(np.where(np.isfinite(x), np.abs(x), 0))))
