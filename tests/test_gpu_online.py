"""GPU parity of the online path (Som::trainSingle / trainBasicSom inner loop) against the
oracle: every transformation x both decay functions, sigma > 1 (full search, window) and
sigma <= 1 (local search, indicator neighbourhood)."""
import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def beq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()
    return (a == b).all()


CASES = [
    ("std_exp", 12, 10, 9, 0, 0, 60, 2.2),
    ("std_inv", 12, 10, 9, 0, 1, 60, 2.2),
    ("std_exp_big_window", 16, 16, 40, 0, 0, 40, 9.0),
    ("median_exp", 9, 9, 17, 1, 0, 50, 1.6),
    ("median_inv", 9, 9, 17, 1, 1, 50, 3.0),
    ("clr_exp", 7, 7, 6, 2, 0, 40, 2.0),
    ("clr_inv", 7, 7, 5, 2, 1, 40, 1.4),
    ("local_sigma1_exp", 10, 10, 8, 0, 0, 60, 1.0),
    ("local_sigma_lt1_inv", 10, 10, 8, 0, 1, 60, 0.7),
    # 2.5 * sigma < 1: the window truncates to nothing (Som.cpp:899-903), the BMU itself is not updated,
    # addBmu / MSE / lastBMU still happen
    ("local_sigma_03_empty_window", 12, 9, 8, 0, 0, 40, 0.3),
    ("local_sigma_03_empty_window_median", 12, 9, 8, 1, 1, 40, 0.3),
    ("d794", 8, 8, 794, 0, 0, 12, 2.5),
    # windows of at most a quarter of the map (written for round 5's look-ahead experiment, profiles/EXPERIMENTS.md, and kept
    # as coverage of bigger maps): chunk sizes that are no multiple of 8, depths with every remainder class of Eigen's
    # reduction (rem >= 4, < 4, < 8 in all)
    ("la_std_exp_rem1", 40, 36, 33, 0, 0, 45, 2.0),
    ("la_std_inv_rem5", 32, 32, 13, 0, 1, 37, 1.8),
    ("la_median_exp_rem4", 30, 30, 20, 1, 0, 50, 2.5),
    ("la_median_inv", 36, 28, 16, 1, 1, 41, 2.2),
    ("la_d794", 48, 48, 794, 0, 0, 21, 3.0),
    ("la_tiny_d3", 24, 24, 3, 0, 0, 30, 1.5),
    ("la_d300_one_block", 40, 40, 300, 0, 1, 8, 2.0),
]


def _run_online_case(W, H, J, tr, fn, B, sigma, X, init, eta, mode=None, reps=2):
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, tr)
    if mode is not None:
        ctx.set_bmu_mode(mode)
    ctx.set_state(map=init)
    for rep in range(reps):          # second pass accumulates weightMap/SMap further (Q9)
        lb = np.zeros(B, np.uint64)
        mse_o = o.train_online_chunk(X, lb, eta, sigma, fn)
        ctx.upload_chunk(X)
        mse_g = ctx.train_online_chunk(eta, sigma, fn)
        got = ctx.get_last_bmu()
        assert beq(got, lb), (rep, "lastBMU", np.nonzero(got != lb)[0][:8], got[got != lb][:8], lb[got != lb][:8])
        st = ctx.get_state()
        for k in ("map", "S", "sigma", "weight", "hits"):
            assert beq(st[k], getattr(o, k)), (rep, k)
        assert beq(np.float32(mse_g), np.float32(mse_o)), (rep, mse_g, mse_o)
    ctx.close()


# The chunk loop's image-bounded search (csrc/vsom_online.hip, "Image-bounded search": sigma > 1, Standard / Median):
# VSOM_BMU_SHORTLIST forces it on every map size, so every case of the table above that it covers runs through it too --
# ragged node counts, depths with every remainder class of Eigen's reduction, windows larger than the map
I8_CASES = [c for c in CASES if c[4] != 2 and c[7] > 1]


@pytest.mark.parametrize("name,W,H,J,tr,fn,B,sigma", I8_CASES, ids=[c[0] for c in I8_CASES])
def test_online_chunk_through_the_image(name, W, H, J, tr, fn, B, sigma):
    X = gen.mnist_like(B, 3, J) if J > 700 else gen.blobs(B, J, 4, 1, 2, sigma=0.3)
    init = gen.random_map(W * H, J, seed=13)
    _run_online_case(W, H, J, tr, fn, B, sigma, X, init, 0.05, mode=capi.BMU_SHORTLIST)


@pytest.mark.parametrize("kind", ["uint8", "unit_floats", "signed_dense", "tiny_values", "huge_values"])
def test_online_image_search_on_a_big_map(kind):
    """A 10 MB map, rows of MNIST-like pixels / the same over 255 / signed dense values / values near the bottom and the
    top of fp32's useful range, through the image-bounded search: the bound has to hold for every scale, and whatever it
    prunes, indices and distances are the exact-order evaluation's."""
    W, H, J, B = 72, 64, 520, 40
    if kind in ("uint8", "unit_floats"):
        X = gen.mnist_like(B, 11, J)
        init = (gen.random_map(W * H, J, seed=5) * np.float32(60) + np.float32(90)).astype(np.float32)
        if kind == "unit_floats":
            X = (X / np.float32(255)).astype(np.float32)
            init = (init / np.float32(255)).astype(np.float32)
    else:
        X = gen.blobs(B, J, 6, 1, 2, sigma=0.5)
        init = gen.random_map(W * H, J, seed=5)
        scale = {"signed_dense": 1.0, "tiny_values": 1e-17, "huge_values": 3e15}[kind]
        X = (X * np.float32(scale)).astype(np.float32)
        init = (init * np.float32(scale)).astype(np.float32)
    _run_online_case(W, H, J, po.STANDARD, capi.EXPONENTIAL, B, 4.0, X, init, 0.08, mode=capi.BMU_SHORTLIST)


def test_online_auto_mode_picks_the_image_at_baseline_size():
    """VSOM_BMU_AUTO on BASELINE's 128 x 128 x 784 map: sigma = 4 goes through the image (vsom_get_online_search_stats counts
    its samples), sigma = 32 -- a window as large as the map -- through the exact scan; both are the oracle's bits, and the
    sigmaMap rows owed at the end of an image chunk are there before vsom_get_state reads them."""
    W = H = 128
    J, B = 784, 64
    X = gen.mnist_like(B, 5, J)
    init = (gen.random_map(W * H, J, seed=42) * np.float32(100) + np.float32(100)).astype(np.float32)
    o = po.OracleSom(W, H, J, po.STANDARD)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, po.STANDARD)
    ctx.set_state(map=init)
    for sigma, through_image in ((4.0, True), (32.0, False), (4.0, True)):
        lb = np.zeros(B, np.uint64)
        mse_o = o.train_online_chunk(X, lb, 0.1, sigma, po.EXPONENTIAL)
        ctx.upload_chunk(X)
        ctx.online_search_stats(reset=True)
        mse_g = ctx.train_online_chunk(0.1, sigma, capi.EXPONENTIAL)
        st = ctx.online_search_stats()
        assert (st["samples"] == B) == through_image, (sigma, st)
        assert beq(ctx.get_last_bmu(), lb), sigma
        state = ctx.get_state()
        for k in ("map", "S", "sigma", "weight", "hits"):
            assert beq(state[k], getattr(o, k)), (sigma, k)
        assert beq(np.float32(mse_g), np.float32(mse_o)), sigma
    ctx.close()


def test_online_image_search_with_nan_inf_rows_and_samples():
    """Rows holding NaN / inf (always candidates, never winners unless the reference says so), duplicates (lowest index), a
    NaN at node 0 (pins the BMU), and samples holding a NaN or an inf (every interval opens: the refinement evaluates all
    nodes exactly) -- through the image-bounded search, both decay functions."""
    W, H, J, B = 36, 30, 24, 43
    X = gen.blobs(B, J, 5, 1, 2, sigma=0.4)
    X[7, 3] = np.nan
    X[19, 0] = np.inf
    init = gen.random_map(W * H, J, seed=21)
    init[500:520] = init[100:120]            # duplicates at higher indices
    init[7, 3] = np.nan
    init[640:700:9] = np.nan
    init[333, 2] = np.inf
    for nan0 in (False, True):
        m = init.copy()
        if nan0:
            m[0, 5] = np.nan
        for fn in (capi.EXPONENTIAL, capi.INVERSE_PROPORTIONAL):
            _run_online_case(W, H, J, po.STANDARD, fn, B, 2.0, X, m, 0.05, mode=capi.BMU_SHORTLIST, reps=1)


def test_online_image_search_one_and_two_sample_chunks():
    """the pipeline's ends: a chunk of one sample (score, refine, window, finish) and of two"""
    W, H, J = 20, 18, 40
    init = gen.random_map(W * H, J, seed=9)
    for B in (1, 2, 3):
        X = gen.blobs(B, J, 3, 1, 4, sigma=0.3)
        _run_online_case(W, H, J, po.MEDIAN, capi.INVERSE_PROPORTIONAL, B, 2.5, X, init, 0.05, mode=capi.BMU_SHORTLIST)


@pytest.mark.parametrize("tr,W,H,J", [(po.STANDARD, 100, 100, 100), (po.STANDARD, 7, 5, 13), (po.MEDIAN, 12, 12, 9),
                                      (po.CLR, 8, 8, 6)])
def test_single_vector_distance_and_local_search(tr, W, H, J):
    """vsom_dist_single / vsom_find_local_bmu (Som::euclidianWeightedDist / Som::findLocalBmu of one host vector, the perf
    harness's million-call scenarios): bit-identical to the oracle; the staged chunk is left alone."""
    D = po.length(tr, J)
    init = gen.random_map(W * H, D, seed=42)
    X = gen.correlated(30, J, seed=5) if tr == po.CLR else gen.blobs(30, J, 4, 1, 2)
    ctx = vsom_amd.Context(W, H, J, tr)
    orc = po.OracleSom(W, H, J, tr)
    ctx.set_state(map=init)
    orc.set_state(map=init)
    ctx.upload_chunk(X[:5])
    rs = np.random.RandomState(3)
    for i in range(len(X)):
        node = int(rs.randint(W * H))
        assert beq(ctx.dist_single(X[i], node), np.float32(orc.dist(node, X[i]))), (i, node)
        start = int(rs.randint(W * H))
        idx, dist = ctx.find_local_bmu(X[i], start)
        want = orc.find_local_bmu(X[i], start)
        assert idx == want, (i, start)
        assert beq(dist, np.float32(orc.dist(want, X[i]))), (i, start)
    assert ctx.chunk_size == 5
    with pytest.raises(capi.VsomError):
        ctx.dist_single(X[0], W * H)
    # the restricted search (Som.cpp:313-332: node 0 seeds whatever its hits) and the all-node distances of findRestrictedBmd
    hits = rs.randint(0, 4, size=W * H).astype(np.uint64)
    ctx.set_state(hits=hits)
    orc.set_state(hits=hits)
    for i in range(12):
        for min_hits in (0, 1, 3, 9):
            idx, dist = ctx.find_restricted_bmu(X[i], min_hits)
            want = orc.find_restricted_bmu(X[i], min_hits)
            assert idx == want, (i, min_hits)
            assert beq(dist, np.float32(orc.dist(want, X[i]))), (i, min_hits)
    all_d = ctx.distances_single(X[3])
    assert beq(all_d, np.array([orc.dist(n, X[3]) for n in range(W * H)], np.float32))
    bad = init.copy()
    bad[0, 0] = np.nan
    ctx.set_state(map=bad)
    orc.set_state(map=bad)
    idx, dist = ctx.find_restricted_bmu(X[0], 2)
    assert idx == orc.find_restricted_bmu(X[0], 2) == 0 and np.isnan(dist)
    assert ctx.chunk_size == 5
    ctx.close()


@pytest.mark.parametrize("name,W,H,J,tr,fn,B,sigma", CASES, ids=[c[0] for c in CASES])
def test_online_chunk(name, W, H, J, tr, fn, B, sigma):
    D = po.length(tr, J)
    X = gen.correlated(B, J, 5) if tr == 2 else (gen.mnist_like(B, 3, J) if J > 700 else gen.blobs(B, J, 4, 1, 2, sigma=0.3))
    init = gen.random_map(W * H, D, seed=13)
    eta = 0.05 if tr != 2 else 0.005
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init)
    for rep in range(2):          # second pass accumulates weightMap/SMap further (Q9)
        lb = np.zeros(B, np.uint64)
        mse_o = o.train_online_chunk(X, lb, eta, sigma, fn)
        ctx.upload_chunk(X)
        mse_g = ctx.train_online_chunk(eta, sigma, fn)
        assert beq(ctx.get_last_bmu(), lb), (name, rep, "lastBMU")
        st = ctx.get_state()
        for k in ("map", "S", "sigma", "weight", "hits"):
            assert beq(st[k], getattr(o, k)), (name, rep, k)
        assert beq(np.float32(mse_g), np.float32(mse_o)), (name, rep, mse_g, mse_o)
    ctx.close()


def test_online_chunk_with_nan_rows_and_duplicates():
    """The online chunk loop on maps that exercise the search's argmin rules: NaN rows never win, equal rows resolve to the
    lowest index, a NaN at node 0 pins every BMU to node 0 (Som.cpp:293-304) -- and the window around node 0 then keeps
    rewriting NaN rows."""
    W, H, J, B = 36, 30, 24, 43
    X = gen.blobs(B, J, 5, 1, 2, sigma=0.4)
    init = gen.random_map(W * H, J, seed=21)
    init[500:520] = init[100:120]            # duplicates at higher indices
    init[7, 3] = np.nan
    init[640:700:9] = np.nan
    for nan0 in (False, True):
        m = init.copy()
        if nan0:
            m[0, 5] = np.nan
        o = po.OracleSom(W, H, J)
        o.set_state(map=m)
        ctx = vsom_amd.Context(W, H, J)
        ctx.set_state(map=m)
        for fn in (capi.EXPONENTIAL, capi.INVERSE_PROPORTIONAL):
            lb = np.zeros(B, np.uint64)
            mse_o = o.train_online_chunk(X, lb, 0.05, 2.0, fn)
            ctx.upload_chunk(X)
            mse_g = ctx.train_online_chunk(0.05, 2.0, fn)
            assert beq(ctx.get_last_bmu(), lb), (nan0, fn)
            assert nan0 == bool((lb == 0).all())
            st = ctx.get_state()
            for k in ("map", "S", "sigma", "weight", "hits"):
                assert beq(st[k], getattr(o, k)), (nan0, fn, k)
            assert beq(np.float32(mse_g), np.float32(mse_o))
        ctx.close()


@pytest.mark.parametrize("tr", [po.STANDARD, po.MEDIAN])
def test_tiny_map_chunk_in_one_launch_with_nan_rows_and_duplicates(tr):
    """Maps of at most 4096 values train a whole chunk in ONE single-workgroup launch (online_tiny_chunk_kernel: model
    values in registers, two barriers per sample, three when sigma <= 1 walks from the sample's last BMU).  The argmin rules on it: NaN rows never win, equal rows resolve to the
    lowest index, a NaN at node 0 pins every BMU to node 0 (Som.cpp:293-304); depths with every remainder class of Eigen's
    reduction; chunks continued with the running MSE; the per-sample forms (fp32 scan, image-bounded search) beside it."""
    for W, H, J, B in ((12, 10, 9, 61), (16, 16, 13, 40), (31, 33, 4, 50), (8, 8, 24, 33), (20, 10, 20, 17)):
        X = gen.blobs(B, J, 5, 1, 2, sigma=0.4)
        init = gen.random_map(W * H, J, seed=21)
        init[50:60] = init[10:20]                # duplicates at higher indices
        init[7, J // 2] = np.nan
        init[33:60:9] = np.nan
        for nan0 in (False, True):
            m = init.copy()
            if nan0:
                m[0, 1] = np.nan
            # VSOM_BMU_AUTO: the one-launch chunk; EXACT / SHORTLIST: the per-sample kernels (fp32 scan / image-bounded search)
            for mode in (capi.BMU_AUTO, capi.BMU_EXACT, capi.BMU_SHORTLIST):
                o = po.OracleSom(W, H, J, tr)
                o.set_state(map=m)
                ctx = vsom_amd.Context(W, H, J, tr)
                ctx.set_bmu_mode(mode)
                ctx.set_state(map=m)
                run_o = np.float32(0)
                # three chunks of full searches, then three of local walks (sigma <= 1, Som.cpp:891) that start from the BMUs
                # the chunk before left in lastBMU; the last one's window is empty (2.5 sigma < 1)
                sched = ((capi.EXPONENTIAL, 2.2), (capi.INVERSE_PROPORTIONAL, 6.0), (capi.EXPONENTIAL, 1.3),
                         (capi.EXPONENTIAL, 1.0), (capi.INVERSE_PROPORTIONAL, 0.8), (capi.EXPONENTIAL, 0.3))
                lb = np.zeros(B, np.uint64)
                for ci, (fn, sigma) in enumerate(sched):
                    lb = lb.copy() if sigma <= 1 else np.zeros(B, np.uint64)
                    start = lb.copy()
                    run_o = o.train_online_chunk(X, lb, 0.05, sigma, fn, mse_start=0.0 if ci == 0 else float(run_o))
                    ctx.upload_chunk(X)
                    ctx.set_last_bmu(start)
                    run_g = ctx.train_online_chunk(0.05, sigma, fn, first_chunk=(ci == 0))
                    assert beq(ctx.get_last_bmu(), lb), (W, H, J, nan0, mode, ci)
                    if sigma > 1:
                        assert nan0 == bool((lb == 0).all())
                    st = ctx.get_state()
                    for k in ("map", "S", "sigma", "weight", "hits"):
                        assert beq(st[k], getattr(o, k)), (W, H, J, nan0, mode, ci, k)
                    assert beq(np.float32(run_g), np.float32(run_o)), (W, H, J, nan0, mode, ci)
                ctx.close()


@pytest.mark.parametrize("W,H,J,B,tr", [(32, 32, 4, 4096, po.STANDARD), (4, 2, 512, 131, po.MEDIAN), (1, 1, 7, 30, po.STANDARD),
                                        (2, 1, 1, 9, po.STANDARD), (32, 16, 8, 4095, po.MEDIAN)])
def test_one_launch_chunk_at_its_limits(W, H, J, B, tr):
    """online_tiny_chunk_kernel at the edges of what it takes: 1024 nodes with the largest chunk (4096 rows: the most LDS it
    asks for), rows of 512 values (four samples per staged block), a single node, one-value rows, a chunk one short of the
    limit with four values per thread; above and at sigma 1."""
    X = gen.blobs(B, J, 3, 1, 2, sigma=0.3)
    init = gen.random_map(W * H, J, seed=4)
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    run_o = np.float32(0)
    for ci, (sigma, fn) in enumerate(((3.0, capi.EXPONENTIAL), (1.0, capi.INVERSE_PROPORTIONAL))):
        lb = lb.copy() if sigma <= 1 else np.zeros(B, np.uint64)
        start = lb.copy()
        run_o = o.train_online_chunk(X, lb, 0.02, sigma, fn, mse_start=0.0 if ci == 0 else float(run_o))
        ctx.upload_chunk(X)
        ctx.set_last_bmu(start)
        run_g, lb_g = ctx.train_online_chunk_fetch(0.02, sigma, fn, first_chunk=(ci == 0))
        assert beq(lb_g, lb) and beq(np.float32(run_g), np.float32(run_o)), ci
        st = ctx.get_state()
        for k in ("map", "S", "sigma", "weight", "hits"):
            assert beq(st[k], getattr(o, k)), (ci, k)
    ctx.close()


@pytest.mark.parametrize("W,H,J,tr", [(10, 10, 9, po.STANDARD), (12, 9, 20, po.MEDIAN), (24, 24, 48, po.STANDARD), (7, 7, 6, po.CLR)])
def test_chunk_with_results_fetched_in_the_same_call(W, H, J, tr):
    """vsom_train_online_chunk_fetch = the chunk's sample loop + its lastBMU + the running MSE in one synchronising call (what
    Som::trainBasicSom reads after the last chunk of an epoch, Som.cpp:895,1163,1167).  On maps of at most 4096 values the
    one-launch kernel stores the results into pinned memory itself; elsewhere the call falls back to the separate getters.
    Two chunks of an epoch (MSE carried), sigma above and at 1, and vsom_get_last_bmu / vsom_get_mse agree afterwards."""
    D = po.length(tr, J)
    B = 37
    X = gen.correlated(B, J, 5) if tr == po.CLR else gen.blobs(B, J, 4, 1, 2, sigma=0.3)
    init = gen.random_map(W * H, D, seed=9)
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init)
    run_o = np.float32(0)
    lb_prev = np.zeros(B, np.uint64)
    for ci, (sigma, fn) in enumerate(((2.5, capi.EXPONENTIAL), (1.0, capi.INVERSE_PROPORTIONAL), (4.0, capi.INVERSE_PROPORTIONAL))):
        lb = lb_prev.copy() if sigma <= 1 else np.zeros(B, np.uint64)
        start = lb.copy()
        run_o = o.train_online_chunk(X, lb, 0.05, sigma, fn, mse_start=0.0 if ci == 0 else float(run_o))
        ctx.upload_chunk(X)
        ctx.set_last_bmu(start)
        run_g, lb_g = ctx.train_online_chunk_fetch(0.05, sigma, fn, first_chunk=(ci == 0))
        assert beq(lb_g, lb), (ci, "lastBMU from the call")
        assert beq(np.float32(run_g), np.float32(run_o)), (ci, run_g, run_o)
        assert beq(ctx.get_last_bmu(), lb) and beq(np.float32(ctx.get_mse()), np.float32(run_o)), ci
        st = ctx.get_state()
        for k in ("map", "S", "sigma", "weight", "hits"):
            assert beq(st[k], getattr(o, k)), (ci, k)
        lb_prev = lb
    ctx.close()


@pytest.mark.parametrize("tr,fn,sigma", [(0, 0, 3.0), (1, 1, 2.0), (2, 0, 1.5), (0, 1, 1.0)])
def test_train_single_api(tr, fn, sigma):
    W, H, J = 11, 9, 6
    D = po.length(tr, J)
    X = gen.correlated(10, J, 5) if tr == 2 else gen.blobs(10, J, 3, 1, 2, sigma=0.3)
    init = gen.random_map(W * H, D, seed=3)
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init)
    last_o = last_g = 5
    for j in range(10):
        bo, ro, do, last_o = o.train_single(X[j], 0.02, sigma, last_o, fn)
        bg, rg, dg, last_g = ctx.train_single(X[j], 0.02, sigma, last_g, fn)
        assert bo == bg and last_o == last_g
        assert beq(ro, rg) and beq(do, dg)
    st = ctx.get_state()
    for k in ("map", "S", "sigma", "weight"):
        assert beq(st[k], getattr(o, k)), k
    assert (st["hits"] == 0).all()      # trainSingle itself does not count hits (addBmu is the driver's)
    ctx.close()


@pytest.mark.parametrize("tr,W,H,J", [(po.STANDARD, 100, 100, 100), (po.STANDARD, 7, 5, 13), (po.MEDIAN, 12, 12, 9),
                                      (po.CLR, 8, 8, 6)])
def test_find_bmu_single_vector(tr, W, H, J):
    """vsom_find_bmu (Som::findBmu of one host vector, the perf harness's findBmu x1000 scenario):
    index and distance bit-identical to the oracle; the staged chunk is left alone."""
    D = po.length(tr, J)
    init = gen.random_map(W * H, D, seed=42)
    X = gen.correlated(40, J, seed=5) if tr == po.CLR else gen.blobs(40, J, 4, 1, 2)
    ctx = vsom_amd.Context(W, H, J, tr)
    orc = po.OracleSom(W, H, J, tr)
    ctx.set_state(map=init)
    orc.set_state(map=init)
    ctx.upload_chunk(X[:7])
    for i in range(len(X)):
        idx, dist = ctx.find_bmu(X[i])
        want = orc.find_bmu(X[i])
        assert idx == want, i
        assert np.float32(dist).view(np.uint32) == np.float32(orc.dist(want, X[i])).view(np.uint32), i
    assert ctx.chunk_size == 7
    # NaN at node 0 pins the BMU to node 0 (Som.cpp:293-299); NaN elsewhere never wins
    bad = init.copy()
    bad[0, 0] = np.nan
    bad[min(5, W * H - 1), 1] = np.nan
    ctx.set_state(map=bad)
    orc.set_state(map=bad)
    idx, dist = ctx.find_bmu(X[0])
    assert idx == orc.find_bmu(X[0]) == 0 and np.isnan(dist)
    bad[0, 0] = init[0, 0]
    ctx.set_state(map=bad)
    orc.set_state(map=bad)
    for i in range(10):
        assert ctx.find_bmu(X[i])[0] == orc.find_bmu(X[i])
    ctx.close()


def test_online_epoch_mse_is_one_running_accumulator():
    """trainBasicSom adds every sample's squaredNorm/epochSize to ONE float declared before the chunk
    loop (Som.cpp:1153,1167): the device keeps that accumulator across chunks."""
    W = H = 9
    J = 11
    X = gen.blobs(90, J, 3, 1, 2)
    init = gen.random_map(W * H, J, seed=42)
    off = [0, 37, 64, 90]
    o = po.OracleSom(W, H, J)
    o.set_state(map=init)
    want = o.train_online(X, off, 1, 0.07, 0.0, 2.5, 0.0, po.EXPONENTIAL)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=init)
    run = np.float32(0)
    sums = []
    for c in range(3):
        ctx.upload_chunk(X[off[c]:off[c + 1]])
        run = ctx.train_online_chunk(0.07, 2.5, capi.EXPONENTIAL, first_chunk=(c == 0))
        sums.append(run)
    got = np.float32(run / np.float32(3))
    assert got.view(np.uint32) == np.float32(want[0]).view(np.uint32)
    # and it is NOT the sum of per-chunk sums in general (different rounding) -- at least it is monotone
    assert sums[0] <= sums[1] <= sums[2]
    ctx.close()
