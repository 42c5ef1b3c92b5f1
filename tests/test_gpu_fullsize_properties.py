"""GPU, BASELINE.json's full size (128x128 map, 784-dim, chunk 4096), where the oracle is too slow
to run end to end: size-independent properties of trainBatchSomEpoch.

 * the new map is algebraically the neighbourhood-weighted mean of the samples (Kohonen Eq. 3.29,
   quoted at Som.cpp:824-834) -> compare with a float64 weighted mean on a random subset of nodes;
 * weightMap[i] = sum_j w(i, bmu_j);  sigmaMap^2 * W = S >= the weighted variance about the final
   mean (each prefix-mean term (x-M_{j-1})^2 dominates West's (x-M_{j-1})(x-M_j));
 * bmuHits grows by exactly B, MSE = mean of the per-sample squared residuals;
 * re-running from the same state reproduces every bit; sample-/node-sharded phases reproduce the
   whole-epoch bits; a random subset of nodes is also checked bit-for-bit against the oracle's phase 2.
"""
import numpy as np
import pytest

import gen
import vsom_amd
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
W = H = 128
J = 784
B = 4096
SIGMA = 32.0


@pytest.fixture(scope="module")
def run():
    X = gen.mnist_like(B, 3, J)
    init = (gen.random_map(W * H, J, 42) * np.float32(100) + np.float32(100)).astype(np.float32)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    mse = ctx.batch_epoch(SIGMA, True)
    st = ctx.get_state()
    out = dict(X=X, init=init, mse=np.float32(mse), st=st, lb=ctx.get_last_bmu(), sq=ctx.get_sqres())
    ctx.close()
    return out


def nbh_matrix(nodes, lb, sigma):
    cx, cy = nodes % W, nodes // W
    bx, by = (lb % W).astype(np.int64), (lb // W).astype(np.int64)
    dx = cx[:, None].astype(np.float64) - bx[None, :]
    dy = cy[:, None].astype(np.float64) - by[None, :]
    return np.exp(-(dx * dx / 2.0 / sigma / sigma + dy * dy / 2.0 / sigma / sigma)).astype(np.float32)


def test_map_is_the_weighted_mean_and_sigma_bounds(run):
    rs = np.random.RandomState(1)
    nodes = rs.choice(W * H, size=384, replace=False)
    wm = nbh_matrix(nodes, run["lb"], SIGMA).astype(np.float64)            # 384 x B
    Wsum = wm.sum(axis=1)
    mean = (wm @ run["X"].astype(np.float64)) / Wsum[:, None]
    got = run["st"]["map"][nodes].astype(np.float64)
    scale = np.abs(mean).max(axis=1, keepdims=True)
    assert (np.abs(got - mean) / scale).max() < 2e-5
    assert np.allclose(run["st"]["weight"][nodes], Wsum, rtol=2e-5)
    # S = sigma^2 * W dominates the weighted variance about the final mean
    S = run["st"]["sigma"][nodes].astype(np.float64) ** 2 * Wsum[:, None]
    x2 = wm @ (run["X"].astype(np.float64) ** 2)
    var = x2 - Wsum[:, None] * mean ** 2                                       # sum w (x - mean)^2
    assert (S >= var * (1 - 1e-4) - 1e-3 * np.abs(x2).max()).all()


def test_counters_and_mse(run):
    assert int(run["st"]["hits"].sum()) == B
    assert (np.bincount(run["lb"].astype(np.int64), minlength=W * H) == run["st"]["hits"]).all()
    m = np.float32(0)
    for q in (run["sq"] / np.float32(B)).astype(np.float32):
        m = np.float32(m + q)
    assert m == run["mse"]
    assert np.isfinite(run["st"]["map"]).all() and np.isfinite(run["st"]["sigma"]).all()


def test_subset_of_nodes_bit_exact_against_oracle_phase2(run):
    o = po.OracleSom(W, H, J)
    o.set_state(map=run["init"])
    for n0 in (0, 5000, 16380):
        n1 = min(n0 + 4, W * H)
        o.batch_phase2_range(run["X"], run["lb"], SIGMA, n0, n1, nthreads=4)
        for k in ("map", "sigma"):
            a, b = run["st"][k][n0:n1], getattr(o, k)[n0:n1]
            assert (a.view(np.uint32) == b.view(np.uint32)).all(), (k, n0)
        assert (run["st"]["weight"][n0:n1].view(np.uint32) == o.weight[n0:n1].view(np.uint32)).all()
    # and the BMUs of a few samples
    o.set_state(map=run["init"])
    for s in (0, 1234, 4095):
        assert o.find_bmu(run["X"][s]) == int(run["lb"][s])


def test_rerun_and_sharded_phases_reproduce_bits(run):
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=run["init"])
    ctx.upload_chunk(run["X"])
    ctx.batch_phase1_async(0, 1500, True)
    ctx.batch_phase1_async(1500, B, True)
    ctx.batch_finish_async()
    ctx.batch_phase2_async(SIGMA, 0, 6000)
    ctx.batch_phase2_async(SIGMA, 6000, W * H)
    st = ctx.get_state()
    assert np.float32(ctx.get_mse()) == run["mse"]
    assert (ctx.get_last_bmu() == run["lb"]).all()
    for k in ("map", "sigma", "weight"):
        assert (st[k].view(np.uint32) == run["st"][k].view(np.uint32)).all(), k
    assert (st["hits"] == run["st"]["hits"]).all()
    ctx.close()
