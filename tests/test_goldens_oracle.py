"""The oracle must keep reproducing the committed golden vectors (tests/golden/*.npz)."""
import glob
import os

import numpy as np
import pytest

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
BATCH = sorted(glob.glob(os.path.join(HERE, "golden", "*batch*.npz")) +
               glob.glob(os.path.join(HERE, "golden", "c4_median.npz")) +
               glob.glob(os.path.join(HERE, "golden", "c5_clr.npz")))
ONLINE = sorted(glob.glob(os.path.join(HERE, "golden", "*online*.npz")))


def beq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()
    return (a == b).all()


@pytest.mark.parametrize("path", BATCH, ids=[os.path.basename(p) for p in BATCH])
def test_batch_goldens(path):
    g = np.load(path)
    W, H, J, tr, epochs = [int(v) for v in g["params"]]
    sigma0, decay = [float(v) for v in g["sched"]]
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=g["init_map"])
    done, mse = o.train_batch(g["X"], g["chunk_off"], epochs, sigma0, decay, nthreads=2)
    assert done == g["map"].shape[0]
    assert beq(o.map, g["map"][-1]) and beq(o.sigma, g["sigma"][-1]) and beq(o.weight, g["weight"][-1])
    assert beq(o.hits, g["hits"]) and beq(mse[:done], g["mse"])


@pytest.mark.parametrize("path", ONLINE, ids=[os.path.basename(p) for p in ONLINE])
def test_online_goldens(path):
    g = np.load(path)
    W, H, J, tr, fn = [int(v) for v in g["params"]]
    eta, sigma = [float(v) for v in g["sched"]]
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=g["init_map"])
    lb = np.zeros(g["X"].shape[0], np.uint64)
    mse = o.train_online_chunk(g["X"], lb, eta, sigma, fn)
    assert beq(lb, g["lastbmu"]) and beq(np.float32(mse), g["mse"])
    for k in ("map", "sigma", "S", "weight", "hits"):
        assert beq(getattr(o, k), g[k]), k
