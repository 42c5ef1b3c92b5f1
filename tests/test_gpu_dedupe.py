"""GPU: the exact search evaluates ONE representative per class of bit-identical model rows (csrc/vsom_bmu.hip,
bmu_row_hash / bmu_row_twin / bmu_unique): equal rows give equal distances and the reference's strict `<` from node 0 keeps
the lowest index (Som.cpp:293-304), so indices and distances must be exactly the oracle's -- on maps where 60 % of the rows
are copies (of earlier AND later rows, of node 0, of rows holding NaN), for each transformation, through the full exact
search and through the redo list of a shortlist search that gives up (a collapsed CLR map)."""
import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def bits_eq(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def map_with_copies(n_nodes, depth, seed, frac=0.6, scale=1.0):
    rs = np.random.RandomState(seed)
    m = (gen.random_map(n_nodes, depth, seed=seed) * np.float32(scale)).astype(np.float32)
    m[11, 3] = np.nan                                   # a NaN row and, below, copies of it (same bits: one class)
    m[17, 0] = np.float32(-0.0)                         # -0 / +0 twins are NOT bit-identical: both stay
    m[18] = m[17]
    m[18, 0] = np.float32(0.0)
    copies = rs.choice(np.arange(1, n_nodes), size=int(frac * n_nodes), replace=False)
    src = rs.randint(0, n_nodes, size=copies.size)      # sources anywhere: earlier and later rows
    for dst, s_ in zip(copies, src):
        if dst not in (11, 17, 18):
            m[dst] = m[s_]
    m[n_nodes - 1] = m[0]                               # node 0's class
    m[n_nodes // 2] = m[11]                             # the NaN row's class
    return m


@pytest.mark.parametrize("tr,W,H,J,B", [(po.STANDARD, 32, 32, 128, 16384), (po.MEDIAN, 40, 28, 120, 16384),
                                        (po.CLR, 16, 16, 24, 16384)])
def test_exact_search_on_a_map_with_duplicate_rows(tr, W, H, J, B):
    D = po.length(tr, J)
    X = gen.correlated(B, J, seed=5) if tr == po.CLR else gen.blobs(B, J, 6, 1, 2, sigma=0.4)
    init = map_with_copies(W * H, D, seed=7)
    orc = po.OracleSom(W, H, J, tr)
    orc.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    sq = np.zeros(B, np.float32)
    orc.batch_phase1_range(X, 0, B, lb, sq, True, nthreads=16)
    for mode in (capi.BMU_EXACT, capi.BMU_AUTO):
        ctx = vsom_amd.Context(W, H, J, tr)
        ctx.set_bmu_mode(mode)
        ctx.set_row_dedupe(0)                           # always (the default waits for searches of 2e10 triples)
        ctx.set_state(map=init)
        ctx.upload_chunk(X)
        idx, dist = ctx.bmu_batch()
        assert (idx == lb).all(), (mode, np.nonzero(idx != lb)[0][:8])
        assert bits_eq(dist, sq), mode
        # NaN at node 0: every BMU is node 0 (Som.cpp:293-299) -- node 0 stays its class's representative
        bad = init.copy()
        bad[0, 1] = np.nan
        ctx.set_state(map=bad)
        idx, dist = ctx.bmu_batch()
        assert (idx == 0).all() and np.isnan(dist).all(), mode
        ctx.close()


def test_redo_list_of_a_shortlist_search_uses_the_representatives():
    """The shortlist hands the samples it cannot bound (here: 300 rows holding a NaN or an inf) to the exact kernel as a
    device-side redo list; with at least 256 entries the list form, too, searches the distinct rows only.  (C5's situation
    is the same path with every sample on the list: a collapsed CLR map.)"""
    W = H = 32
    J, B = 128, 16384
    X = gen.blobs(B, J, 6, 1, 2, sigma=0.4)
    rs = np.random.RandomState(4)
    rows = rs.choice(B, size=300, replace=False)
    X[rows[:200], rs.randint(0, J, size=200)] = np.nan
    X[rows[200:], rs.randint(0, J, size=100)] = np.inf
    init = map_with_copies(W * H, J, seed=12)
    orc = po.OracleSom(W, H, J, po.STANDARD)
    orc.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    sq = np.zeros(B, np.float32)
    orc.batch_phase1_range(X, 0, B, lb, sq, True, nthreads=16)
    ctx = vsom_amd.Context(W, H, J, po.STANDARD)
    ctx.set_bmu_mode(capi.BMU_SHORTLIST)
    ctx.set_row_dedupe(0)                                # always (by default only CLR's redo lists get the passes)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    for rep in range(2):
        idx, dist = ctx.bmu_batch()
        assert (idx == lb).all(), (rep, np.nonzero(idx != lb)[0][:8])
        assert bits_eq(dist, sq), rep
    st = ctx.shortlist_stats()
    ctx.close()
    assert st["redo_samples"] >= 300, st


def test_collapsed_clr_map_goes_through_the_redo_list_with_representatives():
    """C5's situation: a CLR map whose nodes nearly coincide (40 distinct rows that differ in the sixth digit, spread over
    1024 nodes).  The shortlist recognises the collapse on the device, every sample lands on the redo list, and -- by
    default, for the CLR comparer -- the exact kernel then searches the 40 representatives."""
    W = H = 32
    J, B = 20, 4096
    D = po.length(po.CLR, J)
    X = gen.correlated(B, J, seed=9)
    rs = np.random.RandomState(3)
    row = gen.random_map(1, D, seed=11)[0]
    base = np.stack([(row * np.float32(1.0 + 1e-6 * k)).astype(np.float32) for k in range(40)])
    init = base[rs.randint(0, 40, size=W * H)].copy()
    assert len(np.unique(init.view(np.uint32), axis=0)) == 40
    orc = po.OracleSom(W, H, J, po.CLR)
    orc.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    sq = np.zeros(B, np.float32)
    orc.batch_phase1_range(X, 0, B, lb, sq, True, nthreads=16)
    ctx = vsom_amd.Context(W, H, J, po.CLR)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    for rep in range(2):
        idx, dist = ctx.bmu_batch()
        assert (idx == lb).all(), (rep, np.nonzero(idx != lb)[0][:8])
        assert bits_eq(dist, sq), rep
    st = ctx.shortlist_stats()
    ctx.close()
    assert st["redo_samples"] == B, st                   # the collapse was recognised: everything went to the exact kernel
