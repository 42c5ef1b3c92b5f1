"""GPU: the exact retirement of columns that are zero in every row of a chunk (csrc/vsom_compact.hip) -- the
chain kernels and the shortlist contraction run on the live columns only -- must not change one bit of what
Som::trainBatchSomEpoch produces (Som.cpp:756-879): lastBMU, MSE, map, sigmaMap, weightMap, bmuHits against the
oracle (NaN == NaN), on data WITH dead columns and in the corners the argument has to cover:
  * nodes whose first weight underflows (0/0, Som.cpp:857-864): their dead columns are NaN, not 0;
  * NaN / inf model values sitting in a dead column before the search (they still poison / exclude the node);
  * an all-zero chunk, a one-row chunk, -0.0 in an otherwise dead column, non-finite samples;
  * depth 794 (MnistDataLoader's 784 + 10), Median, node shards (group of 3);
  * chunks whose live sets differ, dense chunks in between (the host then skips the passes for a while)."""
import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
THREADS = max(1, min(64, po.max_threads()))


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


def _check_epoch(ctx, orc, X, sigma, first, tag):
    lb = np.zeros(X.shape[0], np.uint64)
    mse_o = orc.batch_epoch(X, lb, sigma, first, nthreads=THREADS)
    ctx.upload_chunk(X)
    mse_g = ctx.batch_epoch(sigma, first)
    assert _same(ctx.get_last_bmu(), lb), (tag, "lastBMU")
    assert _same(np.float32(mse_g), np.float32(mse_o)), (tag, "mse", mse_g, mse_o)
    st = ctx.get_state(S=False)
    for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("weight", orc.weight), ("hits", orc.hits)):
        assert _same(st[k], ref), (tag, k)
    return st


def _pair(W, H, J, tr, init):
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_column_compaction(1)        # the library's default threshold is 1024 rows: these chunks are smaller
    orc = po.OracleSom(W, H, J, tr)
    ctx.set_state(map=init)
    orc.set_state(map=init)
    return ctx, orc


@pytest.mark.parametrize("tr", [po.STANDARD, po.MEDIAN], ids=["standard", "median"])
@pytest.mark.parametrize("W,H,J,B", [(48, 48, 784, 300), (40, 36, 794, 130), (36, 36, 784, 1)],
                         ids=["48x48x784", "40x36x794", "one_row"])
def test_dead_columns_first_and_local_epochs(W, H, J, B, tr):
    X = gen.mnist_like(B, 3, J)
    live, cols = gen.column_occupancy(X)
    assert live + 14 <= cols                      # there IS something to retire
    init = gen.random_map(W * H, J, 42) * np.float32(100)
    ctx, orc = _pair(W, H, J, tr, init)
    _check_epoch(ctx, orc, X, 9.0, True, "first")
    _check_epoch(ctx, orc, X, 7.0, False, "local")
    ctx.close()


def test_poisoned_nodes_get_nan_in_dead_columns():
    """sigma = 2 on a 44x44 map: nodes farther than ~29 cells from the first sample's BMU start with W = 0,
    c_1 = 0/0 -- their whole rows are NaN, dead columns included"""
    W = H = 44
    X = gen.mnist_like(96, 5, 784)
    init = gen.random_map(W * H, 784, 7) * np.float32(100)
    ctx, orc = _pair(W, H, 784, po.STANDARD, init)
    st = _check_epoch(ctx, orc, X, 2.0, True, "poison")
    dead = ~(X != 0).any(axis=0)
    nan_rows = np.isnan(st["map"]).all(axis=1)
    assert nan_rows.any() and not nan_rows.all()
    assert np.isnan(st["map"][nan_rows][:, dead]).all() and (st["map"][~nan_rows][:, dead] == 0).all()
    _check_epoch(ctx, orc, X, 1.6, False, "poison-local")     # the search now walks a map with NaN rows
    ctx.close()


def test_nan_and_inf_model_values_in_dead_columns_reach_the_search():
    W = H = 40
    X = gen.mnist_like(200, 3, 784)
    dead = np.flatnonzero(~(X != 0).any(axis=0))
    init = gen.random_map(W * H, 784, 42) * np.float32(100)
    init[17, dead[0]] = np.nan          # that node can never win (its distance is NaN)
    init[0, dead[1]] = np.nan           # node 0 NaN: Som.cpp:293-299 keeps it as the incumbent
    ctx, orc = _pair(W, H, 784, po.STANDARD, init)
    _check_epoch(ctx, orc, X, 8.0, True, "nan")
    ctx.close()
    init = gen.random_map(W * H, 784, 43) * np.float32(100)
    init[33, dead[2]] = np.inf          # infinite distance: the bound does not apply, the exact kernel redoes
    init[34, dead[3]] = -np.inf
    ctx, orc = _pair(W, H, 784, po.STANDARD, init)
    _check_epoch(ctx, orc, X, 8.0, True, "inf")
    ctx.close()


def test_zero_chunk_negative_zero_and_non_finite_samples():
    W = H = 40
    init = gen.random_map(W * H, 784, 42) * np.float32(100)
    ctx, orc = _pair(W, H, 784, po.STANDARD, init)
    _check_epoch(ctx, orc, np.zeros((50, 784), np.float32), 6.0, True, "all-zero")      # no live column at all
    X = gen.mnist_like(120, 3, 784)
    dead = np.flatnonzero(~(X != 0).any(axis=0))
    X[5, dead[0]] = -0.0                # still a dead column (x == 0), the chain adds -0: M stays +0
    X[9, dead[1]] = np.nan              # NaN != 0: the column is live and the NaN propagates as in the reference
    X[11, dead[2]] = np.inf
    ctx.set_state(map=init, hits=np.zeros(W * H, np.uint64))
    orc.set_state(map=init, hits=np.zeros(W * H, np.uint64))
    _check_epoch(ctx, orc, X, 6.0, True, "odd-values")
    ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_special_values_with_dead_columns_and_zero_blocks(seed):
    """denormals, 3e38, +-inf, NaN and -0 sprinkled over MNIST-like rows (and NaN / inf / denormal model values in
    live AND dead columns): every shortcut -- retired columns, the zero-quad form -- must
    propagate them exactly as the reference's chains do"""
    rs = np.random.RandomState(seed)
    W, H, B = 44, 40, 200 + 37 * seed
    X = gen.mnist_like(B, 10 + seed, 784)
    nz = X != 0
    pick = lambda p: nz & (rs.rand(*X.shape) < p)
    X[pick(0.002)] = np.float32(1e-42)
    X[pick(0.002)] = np.float32(-3e-45)
    X[pick(0.001)] = np.float32(3e38)
    X[pick(0.0005)] = np.inf
    X[pick(0.0005)] = -np.inf
    X[pick(0.0005)] = np.nan
    X[rs.rand(*X.shape) < 0.01] = -0.0
    init = gen.random_map(W * H, 784, 42 + seed) * np.float32(100)
    init[rs.rand(*init.shape) < 0.0005] = np.nan
    init[rs.rand(*init.shape) < 0.0003] = np.inf
    init[rs.rand(*init.shape) < 0.001] = np.float32(2e-41)
    ctx, orc = _pair(W, H, 784, po.STANDARD, init)
    _check_epoch(ctx, orc, X, 7.5, True, "special-first")
    _check_epoch(ctx, orc, X, 5.0, False, "special-local")
    ctx.close()


def test_live_set_changes_between_chunks_and_dense_chunks_pause_the_passes():
    W = H = 40
    J = 784
    init = gen.random_map(W * H, J, 42) * np.float32(100)
    ctx, orc = _pair(W, H, J, po.STANDARD, init)
    rs = np.random.RandomState(1)
    sparse_a = gen.mnist_like(150, 3, J)
    sparse_b = np.roll(gen.mnist_like(90, 4, J).reshape(90, 28, 28), 5, axis=2).reshape(90, J)   # other live set
    dense = (rs.rand(64, J) * 255).astype(np.float32)
    seq = [sparse_a, sparse_b, dense, dense, sparse_a, dense] + [sparse_b] * 10 + [sparse_a]
    for i, X in enumerate(seq):
        _check_epoch(ctx, orc, X, 9.0 - 0.3 * i, i == 0 or i % 3 == 0, f"chunk{i}")
    ctx.close()


def test_default_threshold_and_switch():
    """chunks below 1024 rows are left alone by default, larger ones are compacted; either way and with the
    feature switched off (-1) or forced onto every chunk (1) the results are the oracle's bits"""
    W = H = 40
    init = gen.random_map(W * H, 784, 42) * np.float32(100)
    X = gen.mnist_like(1100, 3, 784)
    for setting in (None, -1, 1):
        ctx = vsom_amd.Context(W, H, 784, po.STANDARD)
        orc = po.OracleSom(W, H, 784, po.STANDARD)
        if setting is not None:
            ctx.set_column_compaction(setting)
        ctx.set_state(map=init)
        orc.set_state(map=init)
        for i, Xc in enumerate((X, X[:300])):
            _check_epoch(ctx, orc, Xc, 9.0, True, f"setting {setting} chunk {i}")
        ctx.close()


@pytest.mark.parametrize("compaction", [1, -1], ids=["compacted", "plain"])
@pytest.mark.parametrize("mode", [capi.UPDATE_STRICT, capi.UPDATE_FMA_SIGMA, capi.UPDATE_FMA], ids=["strict", "sigma", "contracted"])
def test_zero_quad_form_in_every_arithmetic(mode, compaction):
    """~70 % of the (sample, column quad) blocks of an MNIST-like chunk are all zero; the chain kernels then take a
    form without the subtraction (delta = -M; csrc/gen_nt_asm.py, compute_zero) in each of the three arithmetics.
    Against the oracle, with and without the column compaction: strict -- every bit; sigma-contracted -- lastBMU,
    MSE, map, weightMap, bmuHits every bit over the three chunks and sigmaMap within 1e-5 relative;
    contracted -- one epoch from the same map: BMUs / MSE / weightMap / bmuHits every bit, map and sigmaMap within
    1e-5 of max(|reference|, largest sample value of the column) (include/vsom_hip.h, vsom_update_mode)."""
    W = H = 48
    X = gen.mnist_like(2000, 6, 784)
    X[7] = 0.0
    X[100:140, :] = 0.0                                     # a run of all-zero samples
    X[300, 200:260] = -0.0
    init = gen.random_map(W * H, 784, 42) * np.float32(100)
    ctx = vsom_amd.Context(W, H, 784, po.STANDARD)
    ctx.set_column_compaction(compaction)
    ctx.set_update_mode(mode)
    orc = po.OracleSom(W, H, 784, po.STANDARD)
    ctx.set_state(map=init)
    orc.set_state(map=init)

    def close(a, b):
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        nz = b64 != 0
        return bool((np.abs(a64 - b64)[nz] <= 1e-5 * np.abs(b64[nz])).all() and (a64[~nz] == 0).all())

    chunks = ((True, X), (False, X[:1037]), (True, X[5:1994]))      # tails of 5 and 5 samples too
    for i, (first, Xc) in enumerate(chunks[:1] if mode == capi.UPDATE_FMA else chunks):
        lb = np.zeros(Xc.shape[0], np.uint64)
        mse_o = orc.batch_epoch(Xc, lb, 7.0, first, nthreads=THREADS)
        ctx.upload_chunk(Xc)
        mse_g = ctx.batch_epoch(7.0, first)
        st = ctx.get_state(S=False)
        assert _same(ctx.get_last_bmu(), lb) and _same(np.float32(mse_g), np.float32(mse_o)), i
        assert _same(st["weight"], orc.weight) and _same(st["hits"], orc.hits), i
        if mode == capi.UPDATE_STRICT:
            assert _same(st["map"], orc.map) and _same(st["sigma"], orc.sigma), i
        elif mode == capi.UPDATE_FMA_SIGMA:
            assert _same(st["map"], orc.map) and close(st["sigma"], orc.sigma), i
        else:       # the documented bound of the contracted arithmetic (tests/test_gpu_fma_mode.py): 1e-5 of the larger of
            #     the reference element and the magnitude of the operands its chain consumed
            sd = np.abs(Xc).max(axis=0)[None, :].astype(np.float64)
            for k, ref in (("map", orc.map), ("sigma", orc.sigma)):
                a64, b64 = st[k].astype(np.float64), ref.astype(np.float64)
                assert (np.abs(a64 - b64) <= 1e-5 * np.maximum(np.abs(b64), sd)).all(), (i, k)
    ctx.close()


def test_group_of_three_on_data_with_dead_columns():
    """node shards: every member runs the compacted chains on its third of the nodes"""
    W, H, J, B = 48, 45, 784, 240
    X = gen.mnist_like(B, 3, J)
    init = gen.random_map(W * H, J, 42) * np.float32(100)
    orc = po.OracleSom(W, H, J, po.STANDARD)
    orc.set_state(map=init)
    grp = capi.Group(W, H, J, capi.STANDARD, devices=[0, 0, 0])
    for r in range(3):
        grp.member(r).set_column_compaction(1)
    grp.set_state(map=init)
    for e, sigma in enumerate((8.0, 6.5, 2.0)):
        lb = np.zeros(B, np.uint64)
        mse_o = orc.batch_epoch(X, lb, sigma, e == 0, nthreads=THREADS)
        grp.upload_chunk(X)
        if e:
            grp.set_last_bmu(np.zeros(B, np.uint64))
        mse_g = grp.batch_epoch(sigma, e == 0)
        assert _same(grp.get_last_bmu(), lb) and _same(np.float32(mse_g), np.float32(mse_o)), e
        st = grp.get_state()
        for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("weight", orc.weight), ("hits", orc.hits)):
            assert _same(st[k], ref), (e, k)
    grp.close()
