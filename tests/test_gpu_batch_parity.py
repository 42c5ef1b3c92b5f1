"""GPU parity: libvsom_hip.so (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bar (BASELINE.json north_star): BMU indices bit-exact; map / sigmaMap / weightMap
within 1e-5 relative fp32 -- the strict kernels are in fact required to be bit-identical
(NaN == NaN), which is what these tests assert.
"""
import numpy as np
import pytest

import gen
import vsom_amd
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _mk(width, height, J, transform, init_map):
    ctx = vsom_amd.Context(width, height, J, transform)
    orc = po.OracleSom(width, height, J, transform)
    orc.set_state(map=init_map)
    ctx.set_state(map=init_map)
    return ctx, orc


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    if a.dtype.kind == "f":
        ok = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    else:
        ok = a == b
    if not ok.all():
        bad = np.argwhere(~ok)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} / {a.size} differ, first at {i}: gpu={a[i]!r} oracle={b[i]!r}")


def _check_state(ctx, orc, what=""):
    st = ctx.get_state()
    _same(st["map"], orc.map, what + " map")
    _same(st["sigma"], orc.sigma, what + " sigmaMap")
    _same(st["weight"], orc.weight, what + " weightMap")
    _same(st["hits"], orc.hits, what + " bmuHits")


CASES = [
    # name, W, H, J, transform, B, sigma
    ("c1_10x10x16_std", 10, 10, 16, po.STANDARD, 256, 5.0),
    ("fixture_like_9dim", 10, 10, 9, po.STANDARD, 20, 10.0),
    ("ragged_7x5x13", 7, 5, 13, po.STANDARD, 67, 3.0),
    ("d3_small", 6, 6, 3, po.STANDARD, 33, 2.5),
    ("d5", 8, 8, 5, po.STANDARD, 64, 4.0),
    ("d794_mnist_loader", 12, 12, 794, po.STANDARD, 96, 6.0),
    ("c4_median_32", 16, 16, 32, po.MEDIAN, 200, 8.0),
    ("median_ragged", 9, 9, 11, po.MEDIAN, 50, 4.0),
    ("c5_clr_J8", 8, 8, 8, po.CLR, 128, 4.0),
    ("clr_J5_ragged", 6, 7, 5, po.CLR, 45, 3.0),
    # node counts that select the 32- and 64-node role-split neighbourhood kernels, ragged chunks
    ("cw32_96x96x6", 96, 96, 6, po.STANDARD, 333, 20.0),
    ("cw64_128x128x4", 128, 128, 4, po.STANDARD, 77, 30.0),
    ("cw64_130x127x3_median", 130, 127, 3, po.MEDIAN, 130, 25.0),
    # neighbourhood table larger than the LDS budget (200x200x4 B = 160 KB): read from global memory
    ("cw64_lut_global_200x200x3", 200, 200, 3, po.STANDARD, 41, 40.0),
]


@pytest.mark.parametrize("name,W,H,J,tr,B,sigma", CASES, ids=[c[0] for c in CASES])
def test_batch_epoch_first_and_later(name, W, H, J, tr, B, sigma):
    D = po.length(tr, J)
    if tr == po.CLR:
        X = gen.correlated(B, J, seed=5)
    elif J >= 700:
        X = gen.mnist_like(B, seed=3, dim=J)
    else:
        X = gen.blobs(B, J, 4, 1, 2)
    init = gen.random_map(W * H, D, seed=42)
    ctx, orc = _mk(W, H, J, tr, init)

    # epoch 0: findBmu path
    lb = np.zeros(B, np.uint64)
    mse_o = orc.batch_epoch(X, lb, sigma, True)
    ctx.upload_chunk(X)
    mse_g = ctx.batch_epoch(sigma, True)
    _same(ctx.get_last_bmu(), lb, name + " lastBMU(first)")
    _same(np.float32(mse_g), np.float32(mse_o), name + " mse(first)")
    _check_state(ctx, orc, name + " first")

    # epoch 1: the chunk is reloaded (lastBMU zeroed) and findLocalBmu is used
    lb2 = np.zeros(B, np.uint64)
    mse_o2 = orc.batch_epoch(X, lb2, sigma * 0.9, False)
    ctx.upload_chunk(X)
    mse_g2 = ctx.batch_epoch(sigma * 0.9, False)
    _same(ctx.get_last_bmu(), lb2, name + " lastBMU(local)")
    _same(np.float32(mse_g2), np.float32(mse_o2), name + " mse(local)")
    _check_state(ctx, orc, name + " local")
    ctx.close()


def test_bmu_batch_and_distances():
    W = H = 20
    J = 24
    B = 300
    X = gen.blobs(B, J, 6, 11, 12, sigma=0.3)
    init = gen.random_map(W * H, J, seed=7)
    ctx, orc = _mk(W, H, J, po.STANDARD, init)
    ctx.upload_chunk(X)
    idx, dist = ctx.bmu_batch()
    exp_idx = np.array([orc.find_bmu(x) for x in X], np.uint64)
    _same(idx, exp_idx, "findBmu")
    exp_d = np.array([orc.dist(int(i), x) for i, x in zip(exp_idx, X)], np.float32)
    _same(dist, exp_d, "bmu distance")
    rs = np.random.RandomState(0)
    nodes = rs.randint(0, W * H, size=500).astype(np.uint64)
    rows = rs.randint(0, B, size=500).astype(np.uint64)
    d = ctx.distances(nodes, rows)
    exp = np.array([orc.dist(int(n), X[int(r)]) for n, r in zip(nodes, rows)], np.float32)
    _same(d, exp, "euclidianWeightedDist")
    ctx.close()


def test_local_bmu_from_given_start():
    W, H, J, B = 15, 11, 10, 120
    X = gen.blobs(B, J, 5, 21, 22, sigma=0.4)
    # a smooth map so that the hill climb takes several steps
    gx, gy = np.meshgrid(np.arange(W), np.arange(H))
    init = np.zeros((W * H, J), np.float32)
    for d in range(J):
        init[:, d] = (np.sin(0.3 * gx + d) + np.cos(0.2 * gy - d)).reshape(-1) * 0.5
    ctx, orc = _mk(W, H, J, po.STANDARD, init)
    ctx.upload_chunk(X)
    rs = np.random.RandomState(3)
    start = rs.randint(0, W * H, size=B).astype(np.uint64)
    ctx.set_last_bmu(start)
    idx, dist = ctx.bmu_local_batch()
    exp = np.array([orc.find_local_bmu(x, int(s)) for x, s in zip(X, start)], np.uint64)
    _same(idx, exp, "findLocalBmu")
    exp_d = np.array([orc.dist(int(i), x) for i, x in zip(exp, X)], np.float32)
    _same(dist, exp_d, "local bmu distance")
    ctx.close()


def test_weight_underflow_nan_propagation():
    """Q7: sigma so small that (float)exp(...) underflows for far nodes -> 0/0 = NaN rows."""
    W = H = 24
    J = 6
    B = 40
    X = gen.blobs(B, J, 2, 31, 32, sigma=0.05)
    init = gen.random_map(W * H, J, seed=9)
    ctx, orc = _mk(W, H, J, po.STANDARD, init)
    lb = np.zeros(B, np.uint64)
    sigma = 1.05
    mse_o = orc.batch_epoch(X, lb, sigma, True)
    ctx.upload_chunk(X)
    mse_g = ctx.batch_epoch(sigma, True)
    assert np.isnan(orc.map).any(), "test input should produce NaN rows"
    _same(np.float32(mse_g), np.float32(mse_o), "mse")
    _check_state(ctx, orc, "underflow")
    # next epoch on a NaN-poisoned map: NaN distances never win (Q3)
    lb2 = np.zeros(B, np.uint64)
    orc.batch_epoch(X, lb2, sigma, True)
    ctx.upload_chunk(X)
    ctx.batch_epoch(sigma, True)
    _same(ctx.get_last_bmu(), lb2, "lastBMU on NaN map")
    _check_state(ctx, orc, "underflow 2")
    ctx.close()


@pytest.mark.parametrize("tr,J", [(po.STANDARD, 75), (po.STANDARD, 122), (po.MEDIAN, 75), (po.CLR, 6)],
                         ids=["std_rd16_pad", "std_rd14_pad", "median", "clr_rp8_pad"])
def test_ragged_last_slice_keeps_row_padding_zero(tr, J):
    """The assembly update kernels run their last slice over the zero padding of the rows.  With
    0/0 weights (Q7) that padding would turn NaN; it must be zero again when a fresh map is set
    over the NaN one and searched by the MFMA shortlist, which reads rows to the padded length."""
    W = H = 40
    B = 48
    X = gen.blobs(B, J, 2, 31, 32, sigma=0.05)
    orc = po.OracleSom(W, H, J, tr)
    ctx = vsom_amd.Context(W, H, J, tr)
    init = gen.random_map(W * H, orc.depth, seed=9)
    orc.set_state(map=init)
    ctx.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    orc.batch_epoch(X, lb, 1.05, True)
    ctx.upload_chunk(X)
    ctx.batch_epoch(1.05, True)
    assert np.isnan(orc.map).any(), "test input should produce NaN rows"
    _check_state(ctx, orc, "ragged underflow")
    fresh = gen.random_map(W * H, orc.depth, seed=10)
    orc.set_state(map=fresh)
    ctx.set_state(map=fresh)
    ctx.set_bmu_mode(vsom_amd.capi.BMU_SHORTLIST)
    lb2 = np.zeros(B, np.uint64)
    mse_o = orc.batch_epoch(X, lb2, 6.0, True)
    ctx.upload_chunk(X)
    mse_g = ctx.batch_epoch(6.0, True)
    _same(ctx.get_last_bmu(), lb2, "lastBMU after a fresh map over NaN rows")
    _same(np.float32(mse_g), np.float32(mse_o), "mse")
    _check_state(ctx, orc, "fresh map")
    ctx.close()


def test_sharded_phases_equal_whole_epoch():
    """phase1 over sample shards + finish + phase2 over node shards == one whole epoch."""
    W, H, J, B = 12, 12, 20, 150
    X = gen.blobs(B, J, 4, 41, 42)
    init = gen.random_map(W * H, J, seed=5)
    ctx, orc = _mk(W, H, J, po.STANDARD, init)
    lb = np.zeros(B, np.uint64)
    mse_o = orc.batch_epoch(X, lb, 4.0, True)
    ctx.upload_chunk(X)
    ctx.batch_phase1_async(0, 70, True)
    ctx.batch_phase1_async(70, B, True)
    ctx.batch_finish_async()
    ctx.batch_phase2_async(4.0, 0, 50)
    ctx.batch_phase2_async(4.0, 50, W * H)
    _same(np.float32(ctx.get_mse()), np.float32(mse_o), "mse")
    _same(ctx.get_last_bmu(), lb, "lastBMU")
    _check_state(ctx, orc, "sharded")
    ctx.close()


def test_empty_chunk_epoch_rewrites_every_neuron():
    """trainBatchSomEpoch over a chunk of zero rows: no BMUs, MSE 0, and phase 2 still runs -- model
    vectors become zero, sigmaMap sqrt(0/0) = NaN, weightMap 0 (Som.cpp:840-875); bmuHits untouched."""
    W, H, J = 9, 7, 13
    init = gen.random_map(W * H, J, seed=42)
    X = gen.blobs(50, J, 4, 1, 2)
    for tr in (po.STANDARD, po.CLR):
        Jt = 5 if tr == po.CLR else J
        D = po.length(tr, Jt)
        init = gen.random_map(W * H, D, seed=42)
        Xt = X[:, :Jt].copy()
        ctx, orc = _mk(W, H, Jt, tr, init)
        lb = np.zeros(50, np.uint64)
        orc.batch_epoch(Xt, lb, 4.0, True)
        ctx.upload_chunk(Xt)
        ctx.batch_epoch(4.0, True)
        empty = np.zeros((0, Jt), np.float32)
        mse_o = orc.batch_epoch(empty, np.zeros(0, np.uint64), 4.0, False)
        ctx.upload_chunk(empty)
        mse_g = ctx.batch_epoch(4.0, False)
        assert np.float32(mse_g) == np.float32(mse_o) == 0
        assert ctx.chunk_size == 0
        _check_state(ctx, orc, "empty chunk")
        st = ctx.get_state()
        assert (st["map"] == 0).all() and np.isnan(st["sigma"]).all() and (st["weight"] == 0).all()
        ctx.close()
