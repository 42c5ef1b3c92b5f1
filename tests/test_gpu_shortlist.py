"""GPU: the MFMA shortlist BMU search must return exactly what the exact-order search returns
(indices bit-exact, distances bit-identical) -- against the oracle at sizes it finishes in
seconds, and against the exact-order GPU kernel at BASELINE's full size."""
import os

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


def _run(ctx, mode):
    ctx.set_bmu_mode(mode)
    return ctx.bmu_batch()


def _oracle_bmu(o, X, threads=16):
    B = X.shape[0]
    lb = np.zeros(B, np.uint64)
    sq = np.zeros(B, np.float32)
    o.batch_phase1_range(X, 0, B, lb, sq, True, nthreads=threads)
    return lb, sq


CASES = [
    ("mnist_48x48", 48, 48, 784, 384, "mnist"),
    ("d794", 40, 40, 794, 200, "mnist"),
    ("d1024_fp64_epilogue", 36, 36, 1024, 150, "mnist"),      # > 960 contracted columns: the uint8 kind's fp64 epilogue
    ("blobs_64x64x32", 64, 64, 32, 700, "blobs"),
    ("tiny_dim5", 40, 40, 5, 300, "blobs"),
    ("ragged_33x31x77", 33, 31, 77, 150, "blobs"),
    # rows of at most 64 values: the G-less contraction (tile minima only) and its one-wavefront-per-sample refinement
    ("gless_u8_d48_n1295", 37, 35, 48, 333, "mnist"),
    ("gless_d64", 40, 40, 64, 500, "blobs"),
    ("gless_d20_n1023", 33, 31, 20, 150, "blobs"),
    # the G-less ring kernel (>= 256 tiles of 256 x 128) with ragged edges: 9000 nodes = 70.3 tiles, 1000 samples = 3.9
    ("ring_gless_ragged_u8", 100, 90, 300, 1000, "mnist"),
    ("ring_gless_ragged_general", 100, 90, 200, 1000, "blobs"),
]


@pytest.mark.parametrize("name,W,H,J,B,kind", CASES, ids=[c[0] for c in CASES])
def test_shortlist_equals_oracle(name, W, H, J, B, kind):
    X = gen.mnist_like(B, 3, J) if kind == "mnist" else gen.blobs(B, J, 6, 1, 2, sigma=0.5)
    if kind == "mnist":
        init = (gen.random_map(W * H, J, 42) * np.float32(120) + np.float32(110)).astype(np.float32)
    else:
        init = gen.random_map(W * H, J, 42)
    o = po.OracleSom(W, H, J)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    lb_o, sq_o = _oracle_bmu(o, X)
    for mode in (capi.BMU_SHORTLIST, capi.BMU_EXACT):
        idx, dist = _run(ctx, mode)
        assert beq(idx, lb_o), (name, mode)
        assert beq(dist, sq_o), (name, mode)
    # a trained-looking map: run one real epoch, then search again on the smooth map
    lb = np.zeros(B, np.uint64)
    o.batch_epoch(X, lb, max(W, H) / 4.0, True, nthreads=16)
    ctx.set_bmu_mode(capi.BMU_SHORTLIST)
    ctx.batch_epoch(max(W, H) / 4.0, True)
    assert beq(ctx.get_last_bmu(), lb), name
    st = ctx.get_state()
    assert beq(st["map"], o.map) and beq(st["sigma"], o.sigma), name
    lb_o, sq_o = _oracle_bmu(o, X)
    ctx.upload_chunk(X)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert beq(idx, lb_o) and beq(dist, sq_o), name + " trained"
    ctx.close()


def test_ties_duplicates_and_zero_map():
    """Equal nodes: the lowest index must win; an all-zero map/chunk makes every node a
    candidate, which must fall back to the exact kernel (still node 0)."""
    W = H = 40
    J, B = 24, 130
    X = gen.blobs(B, J, 4, 1, 2)
    base = gen.random_map(W * H, J, 5)
    dup = base.copy()
    dup[1000:1100] = base[200:300]       # exact duplicates at higher indices
    dup[37] = base[1500]                 # and one at a lower index than its twin
    o = po.OracleSom(W, H, J)
    o.set_state(map=dup)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=dup)
    ctx.upload_chunk(X)
    lb_o, sq_o = _oracle_bmu(o, X, 8)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert beq(idx, lb_o) and beq(dist, sq_o)
    # zero map
    z = np.zeros_like(dup)
    o.set_state(map=z)
    ctx.set_state(map=z)
    lb_o, sq_o = _oracle_bmu(o, X, 8)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert (lb_o == 0).all()
    assert beq(idx, lb_o) and beq(dist, sq_o)
    # zero samples on a zero map
    ctx.upload_chunk(np.zeros((B, J), np.float32))
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert (idx == 0).all() and (dist == 0).all()
    ctx.close()


def test_nan_rows_and_nan_node0():
    W = H = 36
    J, B = 16, 96
    X = gen.blobs(B, J, 4, 1, 2)
    m = gen.random_map(W * H, J, 6)
    m[5:400:7] = np.nan                  # poisoned rows (Q7) never win
    o = po.OracleSom(W, H, J)
    o.set_state(map=m)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=m)
    ctx.upload_chunk(X)
    lb_o, sq_o = _oracle_bmu(o, X, 8)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert beq(idx, lb_o) and beq(dist, sq_o)
    m[0, 3] = np.nan                     # node 0 NaN pins every BMU to 0 (Q3)
    o.set_state(map=m)
    ctx.set_state(map=m)
    lb_o, sq_o = _oracle_bmu(o, X, 8)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert (lb_o == 0).all() and beq(idx, lb_o) and beq(dist, sq_o)
    # +inf in the map: the bound is not applicable -> device-side fallback to the exact kernel
    m = gen.random_map(W * H, J, 6)
    m[77, 2] = np.inf
    o.set_state(map=m)
    ctx.set_state(map=m)
    lb_o, sq_o = _oracle_bmu(o, X, 8)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert beq(idx, lb_o) and beq(dist, sq_o)
    ctx.close()


def test_full_size_shortlist_equals_exact_kernel():
    """BASELINE size (128x128 map, 784-dim, B=4096): shortlist == exact-order kernel for every
    sample; a random subset is also checked against the oracle."""
    W = H = 128
    J, B = 784, 4096
    X = gen.mnist_like(B, 3, J)
    init = (gen.random_map(W * H, J, 42) * np.float32(100) + np.float32(100)).astype(np.float32)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    i_s, d_s = _run(ctx, capi.BMU_SHORTLIST)
    i_e, d_e = _run(ctx, capi.BMU_EXACT)
    assert beq(i_s, i_e) and beq(d_s, d_e)
    # after one epoch the map is smooth (neighbouring nodes nearly equal): the hard case
    ctx.set_bmu_mode(capi.BMU_EXACT)
    ctx.batch_epoch(32.0, True)
    ctx.upload_chunk(X)
    i_s, d_s = _run(ctx, capi.BMU_SHORTLIST)
    i_e, d_e = _run(ctx, capi.BMU_EXACT)
    assert beq(i_s, i_e) and beq(d_s, d_e)
    st = ctx.get_state(sigma=False, S=False, weight=False, hits=False)
    o = po.OracleSom(W, H, J)
    o.set_state(map=st["map"])
    rs = np.random.RandomState(0)
    for s in rs.randint(0, B, size=24):
        assert o.find_bmu(X[s]) == int(i_s[s])
    ctx.close()


@pytest.mark.parametrize("W,J", [(36, 16), (48, 200)], ids=["small_chain_kernel", "assembly_update"])
def test_nonfinite_and_huge_samples(W, J):
    """NaN, +-inf and 1e30 (squares overflow) inside SAMPLES: the shortlist's bound does not apply to such
    rows (device-side fallback to the exact kernel), every distance of a NaN sample is NaN so its BMU
    is node 0 (Som.cpp:293-309), and phase 2 spreads the NaN through every chain of its column -- all
    as in the oracle, bit for bit."""
    H, B = W, 80
    X = gen.blobs(B, J, 4, 1, 2)
    X[3, 1] = np.nan
    X[17, 0] = np.inf
    X[18, J - 1] = -np.inf
    X[40, 2] = 1e30
    X[41, :] = 3e19                      # squares near FLT_MAX, their sum overflows
    m = gen.random_map(W * H, J, 6)
    o = po.OracleSom(W, H, J)
    o.set_state(map=m)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=m)
    ctx.upload_chunk(X)
    lb_o, sq_o = _oracle_bmu(o, X, 8)
    for mode in (capi.BMU_SHORTLIST, capi.BMU_EXACT, capi.BMU_AUTO):
        idx, dist = _run(ctx, mode)
        assert beq(idx, lb_o) and beq(dist, sq_o), mode
    assert lb_o[3] == 0
    # the whole epoch on those rows (sigma large enough that every node weighs every sample)
    ctx.set_bmu_mode(capi.BMU_AUTO)
    lb = np.zeros(B, np.uint64)
    mse_o = o.batch_epoch(X, lb, 20.0, True)
    ctx.upload_chunk(X)
    mse_g = ctx.batch_epoch(20.0, True)
    assert beq(np.float32(mse_g), np.float32(mse_o))
    st = ctx.get_state()
    for k, ref in (("map", o.map), ("sigma", o.sigma), ("weight", o.weight), ("hits", o.hits)):
        assert beq(st[k], ref), k
    # chains are per (node, dim): the columns holding NaN / +-inf samples are NaN in every node
    assert np.isnan(o.map[:, [0, 1, J - 1]]).all() and not np.isnan(o.map[:, 3]).any()
    ctx.close()


# ---- CombinatorialLinearRegression comparer (Transformation.cpp:82-106): one contraction of length
#      P + 3J over derived features prunes, the exact-order kernel decides (vsom_shortlist.hip) ----
CLR_CASES = [
    ("clr_J24", 32, 32, 24, 300),        # P = 276 (not a multiple of 32)
    ("clr_J12_ragged", 37, 29, 12, 130), # P = 66, non-square map
    ("clr_J64", 32, 32, 64, 256),        # C5's shape: P = 2016
    ("clr_J3", 40, 40, 3, 200),          # P = 3: everything is tail
    ("clr_J33", 33, 33, 33, 96),         # odd J, P = 528
]


def _clr_pair(W, H, J, B, seed=42):
    X = gen.correlated(B, J, 5)
    init = gen.random_map(W * H, capi.model_length(capi.CLR, J), seed)
    o = po.OracleSom(W, H, J, po.CLR)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, capi.CLR)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    return X, init, o, ctx


@pytest.mark.parametrize("name,W,H,J,B", CLR_CASES, ids=[c[0] for c in CLR_CASES])
def test_clr_shortlist_equals_oracle(name, W, H, J, B):
    X, init, o, ctx = _clr_pair(W, H, J, B)
    lb_o, sq_o = _oracle_bmu(o, X)
    for mode in (capi.BMU_SHORTLIST, capi.BMU_EXACT):
        idx, dist = _run(ctx, mode)
        assert beq(idx, lb_o), (name, mode)
        assert beq(dist, sq_o), (name, mode)
    # trained maps (small residuals: the expansion cancels, the bound has to cope): three epochs
    sig = max(W, H) / 4.0
    ctx.set_bmu_mode(capi.BMU_SHORTLIST)
    for ep in range(3):
        lb = np.zeros(B, np.uint64)
        o.batch_epoch(X, lb, sig, ep == 0, nthreads=16)
        ctx.upload_chunk(X)
        ctx.batch_epoch(sig, ep == 0)
        assert beq(ctx.get_last_bmu(), lb), (name, ep)
        lb_o, sq_o = _oracle_bmu(o, X)
        ctx.upload_chunk(X)
        idx, dist = _run(ctx, capi.BMU_SHORTLIST)
        assert beq(idx, lb_o) and beq(dist, sq_o), (name, "trained", ep)
    st = ctx.get_state()
    assert beq(st["map"], o.map) and beq(st["sigma"], o.sigma), name
    stats = ctx.shortlist_stats()
    assert stats["searches"] > 0
    ctx.close()


def test_clr_shortlist_duplicates_zero_map_nan_and_inf_rows():
    W = H = 34
    J, B = 10, 120
    X, init, o, ctx = _clr_pair(W, H, J, B, seed=5)
    m = init.copy()
    m[900:1000] = m[100:200]             # exact duplicates at higher indices: the lowest index wins
    m[17] = m[1100]
    m[5:400:7] = np.nan                  # poisoned rows never win
    for mm in (m, np.zeros_like(m)):
        o.set_state(map=mm)
        ctx.set_state(map=mm)
        lb_o, sq_o = _oracle_bmu(o, X, 8)
        idx, dist = _run(ctx, capi.BMU_SHORTLIST)
        assert beq(idx, lb_o) and beq(dist, sq_o)
    # NaN at node 0 (Som.cpp:293-299: nothing compares below NaN, node 0 stays) and an inf row
    m2 = init.copy()
    m2[0, 3] = np.nan
    m2[77, 1] = np.inf
    o.set_state(map=m2)
    ctx.set_state(map=m2)
    lb_o, sq_o = _oracle_bmu(o, X, 8)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert beq(idx, lb_o) and beq(dist, sq_o)
    # huge samples: the features overflow, every sample goes to the exact kernel
    Xb = X.copy()
    Xb[::3] *= np.float32(1e20)
    o.set_state(map=init)
    ctx.set_state(map=init)
    ctx.upload_chunk(Xb)
    lb_o, sq_o = _oracle_bmu(o, Xb, 8)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert beq(idx, lb_o) and beq(dist, sq_o)
    ctx.close()


def test_integer_contraction_and_its_fallbacks():
    """The shortlist's contraction runs in exact integer arithmetic on the int8 matrix pipe (csrc/vsom_sl_i8.hip): one
    digit per sample value for chunks of small non-negative integers, three for any other chunk; which kind a chunk is
    is a device-side fact the kernels read.  One context sees, in turn: uint8-valued chunks, a chunk with a non-integer /
    a negative / a 256 / a NaN / an inf value (general kind; the NaN / inf SAMPLE is redone exactly), uint8-valued
    chunks again -- and a second context whose model holds huge, tiny, denormal and exactly-zero rows (the digit
    grid's scale is per row).  Indices and distances equal the oracle's bit for bit every time."""
    W, H, J, B = 40, 36, 784, 260
    rs = np.random.RandomState(11)
    init = (gen.random_map(W * H, J, 42) * np.float32(120) + np.float32(110)).astype(np.float32)
    o = po.OracleSom(W, H, J)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_bmu_mode(capi.BMU_SHORTLIST)
    ctx.set_state(map=init)

    def check(X, tag):
        ctx.upload_chunk(X)
        idx, dist = ctx.bmu_batch()
        lb_o, sq_o = _oracle_bmu(o, X)
        assert beq(idx, lb_o) and beq(dist, sq_o), tag

    Xa = gen.mnist_like(B, 3, J)
    Xb = gen.mnist_like(B, 4, J)
    Xb[:, 300:310] = 255.0                                       # the largest value of the grid
    check(Xa, "uint8 a")
    check(Xb, "uint8 b")
    for k, bad in enumerate((0.5, -1.0, 256.0, np.nan, np.inf)):
        Xc = gen.mnist_like(B, 5 + k, J)
        Xc[17, 401] = bad
        c2 = vsom_amd.Context(W, H, J)                           # a fresh context: the integer contraction is tried
        c2.set_bmu_mode(capi.BMU_SHORTLIST)
        c2.set_state(map=init)
        for X, tag in ((Xa, "before"), (Xc, "odd value"), (Xa, "after 1"), (Xb, "after 2")):
            c2.upload_chunk(X)
            idx, dist = c2.bmu_batch()
            lb_o, sq_o = _oracle_bmu(o, X)
            assert beq(idx, lb_o) and beq(dist, sq_o), (bad, tag)
        c2.close()
    check(gen.blobs(B, J, 6, 1, 2, sigma=0.5), "float data")
    check(Xa, "uint8 again")
    ctx.close()

    # rows of very different magnitude, tiny / denormal / zero rows, a NaN row and node 0 NaN
    init2 = init.copy()
    init2[5] *= np.float32(1e6)
    init2[6] *= np.float32(1e-9)
    init2[7] = np.float32(3e-41) * rs.rand(J).astype(np.float32)
    init2[8] = 0.0
    init2[9, 100] = np.nan
    init2[10] *= np.float32(1e15)
    init2[11, ::7] = np.float32(1e-30)
    o2 = po.OracleSom(W, H, J)
    o2.set_state(map=init2)
    c3 = vsom_amd.Context(W, H, J)
    c3.set_bmu_mode(capi.BMU_SHORTLIST)
    c3.set_state(map=init2)
    # consecutive searches in ONE context alternate the two counter sets (the maxima of |M|^2, eps and |M|_1 the
    # pruning bound needs must be those of THIS search: ADVICE r4); dim samples make the |x|^2 part of the bound
    # negligible, so a stale or empty set would prune the true BMU of the rows scaled by 1e6 / 1e15
    Xdim = np.zeros((64, J), np.float32)
    Xdim[np.arange(64), rs.randint(0, J, 64)] = rs.randint(1, 4, 64).astype(np.float32)
    for X in (Xa, Xdim, Xb, Xdim, np.zeros((40, J), np.float32), Xdim / np.float32(3)):
        c3.upload_chunk(X)
        idx, dist = c3.bmu_batch()
        lb_o, sq_o = _oracle_bmu(o2, X)
        assert beq(idx, lb_o) and beq(dist, sq_o)
    init2[0, 3] = np.nan
    o2.set_state(map=init2)
    c3.set_state(map=init2)
    c3.upload_chunk(Xa)
    idx, dist = c3.bmu_batch()
    lb_o, sq_o = _oracle_bmu(o2, Xa)
    assert beq(idx, lb_o) and beq(dist, sq_o)
    c3.close()


def test_general_kind_integer_contraction():
    """Chunks that are NOT uint8-valued take the three-digit form of the integer contraction (csrc/vsom_sl_i8.hip):
    normalised pixels, signed dense rows, rows of very different scale inside one chunk (the digit grid is per sample),
    rows with one huge outlier, denormal / zero rows, rows holding NaN / inf / overflowing values (redone exactly) --
    against a trained-looking map, a map with rows of very different magnitude, and the all-zero map an empty chunk
    leaves (Som.cpp:840-875; every distance equal, node 0 stays: the select kernel's shortcut), also with NaN nodes."""
    W, H, J, B = 40, 36, 784, 300
    rs = np.random.RandomState(5)
    Xs = gen.float_sparse(B, 3, J)
    Xd = gen.float_dense(B, 7, J)
    Xm = Xd.copy()
    Xm[0::5] *= np.float32(1e-6)
    Xm[1::5] *= np.float32(1e5)
    Xm[2::5] *= np.float32(1e-30)
    Xm[3, :] = 0.0
    Xm[8, 17] = np.float32(1e9)                  # one outlier: every other value of the row falls below the grid
    Xm[13, :] = np.float32(1e-42) * rs.randint(0, 9, J).astype(np.float32)   # denormals
    Xbad = Xs.copy()
    Xbad[4, 100] = np.nan
    Xbad[9, 7] = np.inf
    Xbad[11, 300] = -np.inf
    Xbad[14, 5] = np.float32(3e38)
    Xbad[19, :] = np.float32(2e19)               # |x|^2 overflows
    init = (gen.random_map(W * H, J, 42) * np.float32(0.4) + np.float32(0.45)).astype(np.float32)
    smooth = np.repeat(gen.float_sparse(W * H // 4, 9, J), 4, axis=0) * np.float32(0.9) + init * np.float32(0.02)
    wild = init.copy()
    wild[5] *= np.float32(1e6)
    wild[6] *= np.float32(1e-9)
    wild[7] = np.float32(3e-41) * rs.rand(J).astype(np.float32)
    wild[8] = 0.0
    wild[9, 100] = np.nan
    wild[11, ::7] = np.float32(-1e-30)
    zero = np.zeros_like(init)
    zero_nan = zero.copy()
    zero_nan[3, 2] = np.nan
    zero_nan0 = zero.copy()
    zero_nan0[0, 0] = np.nan
    almost = zero.copy()
    almost[700, 3] = np.float32(1e-44)           # NOT an all-zero map: the shortcut must not fire
    negz = -zero
    o = po.OracleSom(W, H, J)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_bmu_mode(capi.BMU_SHORTLIST)
    for mname, m in (("init", init), ("smooth", smooth.astype(np.float32)), ("wild", wild), ("zero", zero),
                     ("zero+nan", zero_nan), ("zero, node 0 nan", zero_nan0), ("almost zero", almost), ("-0", negz)):
        o.set_state(map=m)
        ctx.set_state(map=m)
        for xname, X in (("sparse", Xs), ("dense", Xd), ("mixed scales", Xm), ("non-finite", Xbad),
                         ("uint8", gen.mnist_like(B, 3, J)), ("tiny", Xs * np.float32(1e-20))):
            ctx.upload_chunk(X)
            idx, dist = ctx.bmu_batch()
            lb_o, sq_o = _oracle_bmu(o, X)
            assert beq(idx, lb_o) and beq(dist, sq_o), (mname, xname)
    # the pruning does prune on a realistic map (the contraction is not silently handing everything to the exact kernel)
    o.set_state(map=smooth.astype(np.float32))
    ctx.set_state(map=smooth.astype(np.float32))
    ctx.upload_chunk(Xs)
    ctx.bmu_batch()
    st = ctx.shortlist_stats()
    assert st["redo_samples"] == 0 and st["candidates"] < 64 * B, st
    ctx.close()


@pytest.mark.parametrize("kind", ["blobs", "u8"])
def test_c4_size_short_rows_search_equals_exact_kernel(kind):
    """BASELINE config 4's search (64x64 map, 32-dim, B = 16384): in the automatic mode rows of at most 64 values go
    through the G-less integer contraction; every index and distance equals the exact-order kernel's, on the random and on
    the trained (smooth) map, and a subset is checked against the oracle."""
    W = H = 64
    J, B = 32, 16384
    if kind == "u8":
        X = gen.mnist_like(B, 5, J)
        init = (gen.random_map(W * H, J, 42) * np.float32(100) + np.float32(100)).astype(np.float32)
    else:
        X = gen.blobs(B, J, 8, 1, 2, sigma=0.5)
        init = gen.random_map(W * H, J, 42)
    ctx = vsom_amd.Context(W, H, J, po.MEDIAN)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    for rnd in range(2):
        i_a, d_a = _run(ctx, capi.BMU_AUTO)
        st = ctx.shortlist_stats()
        i_e, d_e = _run(ctx, capi.BMU_EXACT)
        assert beq(i_a, i_e) and beq(d_a, d_e), (kind, rnd)
        assert st["samples"] == B, st                    # the shortlist path ran
        # (one Median epoch at sigma = 16 leaves a nearly constant map: on uint8 data most nodes then tie within the
        # bound and the samples go to the exact kernel -- correct, just not fast)
        if rnd == 0 or kind == "blobs":
            assert st["redo_samples"] * 100 <= B and st["candidates"] < 256 * B, (rnd, st)
        if rnd == 0:
            ctx.set_bmu_mode(capi.BMU_AUTO)
            ctx.batch_epoch(16.0, True)
            ctx.upload_chunk(X)
    mp = ctx.get_state(sigma=False, S=False, weight=False, hits=False)["map"]
    o = po.OracleSom(W, H, J, po.MEDIAN)
    o.set_state(map=mp)
    rs = np.random.RandomState(1)
    for s_ in rs.randint(0, B, size=32):
        assert o.find_bmu(X[s_]) == int(i_a[s_])
    ctx.close()


def _sl_sweep_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        W, H = int(rs.randint(6, 72)), int(rs.randint(6, 72))
        J = int(rs.choice([rs.randint(2, 65), rs.randint(2, 65), rs.randint(65, 140)]))      # mostly the G-less form (<= 64)
        B = int(rs.choice([rs.randint(1, 130), rs.randint(130, 700)]))
        if i % 6 == 5:       # large enough for the G-less ring kernel (>= 256 tiles of 256 x 128), ragged edges
            W, H = int(rs.randint(90, 129)), int(rs.randint(90, 129))
            J = int(rs.randint(65, 400))
            B = int(rs.randint(600, 1500))
        kind = str(rs.choice(["u8", "u8_sparse", "float", "float_sparse", "tiny", "huge", "mixed_scales"]))
        mapk = str(rs.choice(["random", "trained", "duplicates", "offset"]))
        out.append((f"sl{i}_{W}x{H}x{J}_B{B}_{kind}_{mapk}", W, H, J, B, kind, mapk, int(rs.randint(1, 1 << 30))))
    return out


SL_SWEEP = _sl_sweep_cases(int(os.environ.get("VSOM_SL_SWEEP_N", "24")), int(os.environ.get("VSOM_SWEEP_SEED", "20240611")))


@pytest.mark.parametrize("name,W,H,J,B,kind,mapk,seed", SL_SWEEP, ids=[c[0] for c in SL_SWEEP])
def test_random_shortlist_search_equals_exact_kernel(name, W, H, J, B, kind, mapk, seed):
    """Random shapes, data kinds (uint8 pixels, floats, magnitudes from 1e-20 to 1e15, rows of mixed scale, exact zeros) and
    maps (random, after one epoch, with duplicate rows, far from the data): the forced shortlist search -- the G-less
    form for rows of at most 64 values, the G form above -- returns the exact-order kernel's indices and distances bit for
    bit (the exact kernel is checked against the oracle by every other test).  VSOM_SL_SWEEP_N widens the sweep."""
    rs = np.random.RandomState(seed)
    if kind.startswith("u8"):
        X = rs.randint(0, 256, size=(B, J)).astype(np.float32)
    else:
        X = rs.randn(B, J).astype(np.float32)
    if kind.endswith("sparse"):
        X[rs.rand(B, J) < 0.7] = 0.0
    if kind == "tiny":
        X *= np.float32(1e-20)
    elif kind == "huge":
        X *= np.float32(1e15)
    elif kind == "mixed_scales":
        X *= np.exp(rs.uniform(-30, 30, size=(B, 1))).astype(np.float32)
    scale = np.float32(np.abs(X).max() if np.abs(X).max() > 0 else 1.0)
    init = (gen.random_map(W * H, J, seed % 1000) * scale).astype(np.float32)
    if mapk == "duplicates":
        k = max(1, (W * H) // 5)
        init[-k:] = init[:k]
        init[0] = init[W * H // 2]
    elif mapk == "offset":
        init = (init + np.float32(3) * scale).astype(np.float32)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    if mapk == "trained":
        ctx.set_bmu_mode(capi.BMU_EXACT)
        ctx.batch_epoch(max(W, H) / 5.0, True)
        ctx.upload_chunk(X)
    i_s, d_s = _run(ctx, capi.BMU_SHORTLIST)
    i_e, d_e = _run(ctx, capi.BMU_EXACT)
    bad = np.nonzero(i_s != i_e)[0]
    assert bad.size == 0, (name, bad[:8], i_s[bad[:8]], i_e[bad[:8]])
    assert beq(d_s, d_e), name
    ctx.close()


@pytest.mark.parametrize("kind", ["float", "u8"])
def test_short_rows_search_with_retired_columns(kind):
    """Rows of exactly 64 values in a chunk large enough for the column compaction (>= 1024 rows) with all-zero columns:
    the G-less contraction then runs on the gathered live columns (the 16-lanes-per-row quantisation kernels with the
    live-column list).  Forced shortlist search and a whole epoch against the oracle."""
    W = H = 40
    J, B = 64, 1100
    rs = np.random.RandomState(5)
    if kind == "u8":
        X = rs.randint(0, 256, size=(B, J)).astype(np.float32)
        init = (gen.random_map(W * H, J, 42) * np.float32(100) + np.float32(100)).astype(np.float32)
    else:
        X = gen.blobs(B, J, 6, 1, 2, sigma=0.5)
        init = gen.random_map(W * H, J, 42)
    X[:, [0, 1, 2, 17, 18, 40, 41, 42, 43, 63]] = 0.0        # retired columns (one whole quad among them)
    o = po.OracleSom(W, H, J)
    o.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    lb_o, sq_o = _oracle_bmu(o, X)
    for mode in (capi.BMU_SHORTLIST, capi.BMU_EXACT):
        idx, dist = _run(ctx, mode)
        assert beq(idx, lb_o) and beq(dist, sq_o), (kind, mode)
    lb = np.zeros(B, np.uint64)
    o.batch_epoch(X, lb, 10.0, True, nthreads=16)
    ctx.set_bmu_mode(capi.BMU_SHORTLIST)
    ctx.batch_epoch(10.0, True)
    assert beq(ctx.get_last_bmu(), lb), kind
    st = ctx.get_state()
    assert beq(st["map"], o.map) and beq(st["sigma"], o.sigma), kind
    lb_o, sq_o = _oracle_bmu(o, X)
    ctx.upload_chunk(X)
    idx, dist = _run(ctx, capi.BMU_SHORTLIST)
    assert beq(idx, lb_o) and beq(dist, sq_o), kind + " trained"
    ctx.close()


@pytest.mark.parametrize("kind", ["u8", "float"])
def test_sharded_search_ranges_equal_the_whole_search(kind):
    """A rank of the strong split searches samples [s0, s1) with s0 > 0 (vsom_batch_phase1_async): at C3's map size three
    ragged ranges (each large enough for the G-less ring kernel) must give the indices and distances of one search over the
    whole chunk and of the exact-order kernel."""
    W = H = 128
    J, B = 784, 2048
    X = gen.mnist_like(B, 3, J)
    if kind == "float":
        X = (X / np.float32(255)).astype(np.float32)
        init = gen.random_map(W * H, J, 42).astype(np.float32)
    else:
        init = (gen.random_map(W * H, J, 42) * np.float32(100) + np.float32(100)).astype(np.float32)
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    i_e, d_e = _run(ctx, capi.BMU_EXACT)
    ctx.set_bmu_mode(capi.BMU_AUTO)
    ctx.upload_chunk(X)
    ctx.batch_phase1_async(0, B, True)
    ctx.synchronize()
    i_w, d_w = ctx.get_last_bmu(), ctx.get_sqres()
    ctx.upload_chunk(X)
    for a, b in ((0, 700), (700, 1500), (1500, B)):
        ctx.batch_phase1_async(a, b, True)
    ctx.synchronize()
    i_r, d_r = ctx.get_last_bmu(), ctx.get_sqres()
    assert beq(i_w, i_e) and beq(d_w, d_e), kind
    assert beq(i_r, i_e) and beq(d_r, d_e), kind
    ctx.close()
