"""GPU, C ABI: the device kernels behind the SURVEY 8f "next" rows -- restricted BMU search,
all-node distances of a sample, sigma-normalised raw distances (U-matrix) -- bit for bit against the
oracle."""
import numpy as np
import pytest

import gen
import vsom_amd
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,J,tr", [(14, 11, 13, 0), (9, 9, 6, 2), (20, 20, 784, 0)])
def test_restricted_bmu_rowdist_rawdist(W, H, J, tr):
    D = po.length(tr, J)
    B = 60
    X = gen.correlated(B, J, 5) if tr == 2 else (gen.mnist_like(B, 3, J) if J > 700 else gen.blobs(B, J, 4, 1, 2, sigma=0.4))
    init = gen.random_map(W * H, D, 17) * (np.float32(100) if J > 700 else np.float32(1))
    rs = np.random.RandomState(2)
    hits = rs.randint(0, 6, size=W * H).astype(np.uint64)
    sigma = (rs.rand(W * H, D) * 0.5).astype(np.float32)
    sigma[rs.rand(W * H, D) < 0.1] = 0.0                    # exercises the 1e-5 floor
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init, sigma=sigma, hits=hits)
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init, sigma=sigma, hits=hits)
    ctx.upload_chunk(X)
    for mh in (0, 2, 5, 100):
        idx, dist = ctx.bmu_restricted_batch(mh)
        exp = np.array([o.find_restricted_bmu(x, mh) for x in X], np.uint64)
        assert (idx == exp).all(), mh
        expd = np.array([o.dist(int(i), x) for i, x in zip(exp, X)], np.float32)
        assert (dist.view(np.uint32) == expd.view(np.uint32)).all()
    d = ctx.distances_row(7)
    expd = np.array([o.dist(n, X[7]) for n in range(W * H)], np.float32)
    assert (d.view(np.uint32) == expd.view(np.uint32)).all()
    nodes = rs.randint(0, W * H, size=200).astype(np.uint64)
    nbrs = rs.randint(0, W * H, size=200).astype(np.uint64)
    raw = ctx.distances_raw(nodes, nbrs, True)
    exp = np.array([o.dist_raw(int(n), o.map[int(m)]) for n, m in zip(nodes, nbrs)], np.float32)
    assert (raw.view(np.uint32) == exp.view(np.uint32)).all()
    if tr != 2:
        rows = rs.randint(0, B, size=100).astype(np.uint64)
        raw = ctx.distances_raw(nodes[:100], rows, False)
        exp = np.array([o.dist_raw(int(n), X[int(r)]) for n, r in zip(nodes[:100], rows)], np.float32)
        assert (raw.view(np.uint32) == exp.view(np.uint32)).all()
    ctx.close()


@pytest.mark.parametrize("W,H,J,tr", [(9, 7, 13, 0), (6, 5, 5, 2), (12, 12, 32, 1)])
def test_python_mirror_update_umatrix(W, H, J, tr):
    """Som::updateUMatrix (Som.cpp:999-1111) of the Python mirror: raw distances on the device, the 3/5/8
    neighbour combination on the host, against the oracle's restatement -- also after every epoch of
    trainBatchSom(updateUMatrixAfterEpoch=True) (:751-752)"""
    from vsom_amd import som as vs
    D = po.length(tr, J)
    X = gen.correlated(120, J, 5) if tr == 2 else gen.blobs(120, J, 4, 1, 2, sigma=0.4)
    init = gen.random_map(W * H, D, 17)
    rs = np.random.RandomState(4)
    sigma = (rs.rand(W * H, D) * 0.5).astype(np.float32)
    sigma[rs.rand(W * H, D) < 0.1] = 0.0
    t = vs.Transformation(tr)
    s = vs.Som(W, H, D if tr != 2 else D, t)
    s.setState(map=init, sigma=sigma)
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init, sigma=sigma)
    assert (s.updateUMatrix() == o.update_umatrix()).all()
    ds = vs.ArrayDataSet(X, 60)
    s.trainBatchSom(ds, 2, 3.0, 0.1, updateUMatrixAfterEpoch=True)
    o.train_batch(X, [0, 60, 120], 2, 3.0, 0.1, nthreads=4)
    um, uo = s.getUMatrix(), o.update_umatrix()
    assert ((um == uo) | (np.isnan(um) & np.isnan(uo))).all()
    s.close()
