"""GPU: the exact search on SHORT vectors (part length <= 32 = ONE K-chunk of bmu_tile_kernel, csrc/vsom_bmu.hip: the whole
distance is formed by the remainder / reduction-tree code after at most four 8-blocks; maps like BASELINE config 4).
Som::findBmu (Som.cpp:291-309) and findRestrictedBmu (:311-333) against the oracle, bit for bit: every remainder
class of Eigen's reduction (L mod 8 = 0, < 4, = 4, > 4), node counts that are no multiple of the 64-node tile, chunk sizes
that are no multiple of the 64-sample tile, NaN at node 0 / in a sample, and whole batch epochs (Standard, Median) on
such a map (tests/test_gpu_baseline_configs.py runs C4 itself)."""
import numpy as np
import pytest

import gen
import vsom_amd
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("J", [1, 3, 4, 7, 8, 12, 13, 16, 23, 31, 32])
@pytest.mark.parametrize("W,H,B", [(23, 17, 600), (64, 64, 2049), (5, 5, 513)])
def test_find_bmu_short_vectors(W, H, B, J):
    X = gen.blobs(B, J, 5, 1, 2, sigma=0.3)
    init = gen.random_map(W * H, J, seed=7 + J)
    o = po.OracleSom(W, H, J, po.STANDARD)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, po.STANDARD)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    idx, dist = ctx.bmu_batch()
    step = max(1, B // 150)                       # the oracle's scalar search on a spread of the samples + the tails
    rows = sorted(set(range(0, B, step)) | set(range(B - 70, B)) | set(range(0, 70)))
    exp = np.array([o.find_bmu(X[r]) for r in rows], np.uint64)
    assert (idx[rows] == exp).all(), (W, H, B, J)
    expd = np.array([o.dist(int(i), X[r]) for i, r in zip(exp, rows)], np.float32)
    assert (_bits(dist[rows]) == _bits(expd)).all()
    ctx.close()


def test_restricted_search_and_nan_rules():
    W, H, J, B = 20, 13, 13, 700
    X = gen.blobs(B, J, 4, 1, 2, sigma=0.4)
    X[5, 3] = np.nan                              # a NaN sample: every comparison false, BMU stays node 0
    init = gen.random_map(W * H, J, seed=3)
    rs = np.random.RandomState(2)
    hits = rs.randint(0, 6, size=W * H).astype(np.uint64)
    for nan_node0 in (False, True):
        m = init.copy()
        if nan_node0:
            m[0, 2] = np.nan                      # Som.cpp:293-299: a NaN at node 0 pins every BMU to 0
        o = po.OracleSom(W, H, J, po.STANDARD)
        o.set_state(map=m, hits=hits)
        ctx = vsom_amd.Context(W, H, J, po.STANDARD)
        ctx.set_state(map=m, hits=hits)
        ctx.upload_chunk(X)
        idx, dist = ctx.bmu_batch()
        exp = np.array([o.find_bmu(x) for x in X], np.uint64)
        assert (idx == exp).all(), nan_node0
        for mh in (0, 2, 5, 100):
            idx, dist = ctx.bmu_restricted_batch(mh)
            exp = np.array([o.find_restricted_bmu(x, mh) for x in X], np.uint64)
            assert (idx == exp).all(), (nan_node0, mh)
            expd = np.array([o.dist(int(i), x) for i, x in zip(exp, X)], np.float32)
            ok = (_bits(dist) == _bits(expd)) | (np.isnan(dist) & np.isnan(expd))
            assert ok.all(), (nan_node0, mh)
        ctx.close()


@pytest.mark.parametrize("tr", [po.STANDARD, po.MEDIAN], ids=["standard", "median"])
def test_batch_epochs_on_short_vectors(tr):
    W, H, J, B = 40, 36, 20, 3000
    X = gen.blobs(B, J, 6, 3, 4)
    init = gen.random_map(W * H, J, seed=11)
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init)
    for e, s in enumerate((9.0, 7.0)):
        lb = np.zeros(B, np.uint64)
        mse_o = o.batch_epoch(X, lb, s, e == 0, nthreads=max(1, min(64, po.max_threads())))
        ctx.upload_chunk(X)
        mse_g = ctx.batch_epoch(s, e == 0)
        assert (ctx.get_last_bmu() == lb).all(), e
        assert _bits(mse_g) == _bits(mse_o), e
        st = ctx.get_state()
        for k, ref in (("map", o.map), ("sigma", o.sigma), ("weight", o.weight)):
            a, b = _bits(st[k]), _bits(ref)
            assert ((a == b) | (np.isnan(st[k]) & np.isnan(ref))).all(), (e, k)
        assert (st["hits"] == o.hits).all()
    ctx.close()
