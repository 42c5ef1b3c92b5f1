"""The C oracle against the independent pure-Python restatement (tests/pyref.py), bit for bit,
on small seeded cases covering every transformation, ragged lengths (Eigen tail handling),
non-square maps (SomIndex quirk), NaN propagation and both online decay functions."""
import numpy as np
import pytest

import gen
import pyref
from oracle import pyoracle as po


def _bits_equal(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


@pytest.mark.parametrize("n", list(range(0, 41)) + [63, 64, 65, 784, 794, 2016])
def test_dot_self_order(n):
    rs = np.random.RandomState(n)
    r = (rs.randn(n) * 10).astype(np.float32)
    assert _bits_equal(po.dot_self(r), pyref.dot_self(r))


@pytest.mark.parametrize("tr,J", [(0, 5), (1, 6), (2, 4), (2, 5)])
def test_comparer_stepper(tr, J):
    rs = np.random.RandomState(J)
    v = rs.randn(J).astype(np.float32)
    m = rs.randn(pyref.length(tr, J)).astype(np.float32)
    assert _bits_equal(po.comparer(tr, v, m), pyref.comparer(tr, v, m))
    assert _bits_equal(po.stepper(tr, v, m), pyref.stepper(tr, v, m))


def test_median_sign_zero_and_nan():
    v = np.array([1.0, 2.0, np.nan, -0.0], np.float32)
    m = np.array([1.0, 3.0, 0.0, 0.0], np.float32)
    s = po.stepper(po.MEDIAN, v, m)
    assert s[0] == 0 and not np.signbit(s[0]) and s[1] == -1 and np.isnan(s[2]) and s[3] == 0


BATCH = [
    ("std_sq", 5, 5, 7, 0, 23, 2.5),
    ("std_nonsquare", 6, 4, 9, 0, 17, 3.0),
    ("std_nonsquare_tall", 3, 7, 4, 0, 15, 2.0),
    ("median", 4, 4, 5, 1, 19, 2.0),
    ("clr", 4, 3, 4, 2, 13, 2.0),
    ("underflow_nan", 14, 14, 3, 0, 9, 1.02),
    ("sigma_le_1", 4, 4, 3, 0, 11, 1.0),
]


@pytest.mark.parametrize("name,W,H,J,tr,B,sigma", BATCH, ids=[b[0] for b in BATCH])
def test_batch_epochs(name, W, H, J, tr, B, sigma):
    X = gen.correlated(B, J, 5) if tr == 2 else gen.blobs(B, J, 3, 1, 2, sigma=0.3)
    init = gen.random_map(W * H, pyref.length(tr, J), seed=11)
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    p = pyref.Som(W, H, J, tr)
    p.map[...] = init
    for ep, first in enumerate((True, False, False)):
        lo, lp = np.zeros(B, np.uint64), np.zeros(B, np.uint64)
        mo = o.batch_epoch(X, lo, sigma, first)
        mp = p.batch_epoch(X, lp, sigma, first)
        assert (lo == lp).all(), (name, ep)
        assert _bits_equal(mo, mp), (name, ep, mo, mp)
        assert _bits_equal(o.map, p.map) and _bits_equal(o.sigma, p.sigma), (name, ep)
        assert _bits_equal(o.weight, p.weight) and (o.hits == p.hits).all(), (name, ep)
    if name == "underflow_nan":
        assert np.isnan(o.map).any()


def test_threads_and_faithful_variants_agree():
    W, H, J, B = 9, 8, 12, 40
    X = gen.blobs(B, J, 3, 1, 2)
    init = gen.random_map(W * H, J, seed=3)
    outs = []
    for kw in ({"nthreads": 1}, {"nthreads": 4}, {"faithful": True}):
        o = po.OracleSom(W, H, J)
        o.set_state(map=init)
        lb = np.zeros(B, np.uint64)
        mse = o.batch_epoch(X, lb, 3.0, True, **kw)
        outs.append((lb.copy(), mse, o.map.copy(), o.sigma.copy(), o.weight.copy()))
    for other in outs[1:]:
        assert (outs[0][0] == other[0]).all() and _bits_equal(outs[0][1], other[1])
        for a, b in zip(outs[0][2:], other[2:]):
            assert _bits_equal(a, b)


@pytest.mark.parametrize("tr,fn,sigma", [(0, 0, 2.0), (0, 1, 2.0), (1, 0, 1.5), (2, 1, 1.7),
                                         (0, 0, 1.0), (0, 1, 0.8)])
def test_train_single_sequence(tr, fn, sigma):
    W, H, J, B = 7, 6, 4, 12
    X = gen.correlated(B, J, 5) if tr == 2 else gen.blobs(B, J, 3, 1, 2, sigma=0.3)
    init = gen.random_map(W * H, pyref.length(tr, J), seed=5)
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    p = pyref.Som(W, H, J, tr)
    p.map[...] = init
    last_o = last_p = 0
    for j in range(B):
        bo, ro, do, last_o = o.train_single(X[j], 0.1, sigma, last_o, fn)
        bp, rp, dp, last_p = p.train_single(X[j], 0.1, sigma, last_p, fn)
        assert bo == bp and last_o == last_p
        assert _bits_equal(ro, rp) and _bits_equal(do, dp)
    assert _bits_equal(o.map, p.map) and _bits_equal(o.S, p.S)
    assert _bits_equal(o.sigma, p.sigma) and _bits_equal(o.weight, p.weight)


def test_find_local_bmu_random_starts():
    W, H, J = 9, 7, 5
    init = gen.random_map(W * H, J, seed=8)
    o = po.OracleSom(W, H, J)
    o.set_state(map=init)
    p = pyref.Som(W, H, J)
    p.map[...] = init
    rs = np.random.RandomState(1)
    for _ in range(60):
        v = rs.randn(J).astype(np.float32) * 0.5
        st = int(rs.randint(0, W * H))
        assert o.find_local_bmu(v, st) == p.find_local_bmu(v, st)
        assert o.find_bmu(v) == p.find_bmu(v)
