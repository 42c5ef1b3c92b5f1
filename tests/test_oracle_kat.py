"""Known answers derived by hand from the reference source and the
reference's own test data (tests/performance/data/testDb.sq3 rows; the CLR pair order printed
by tests/test1.cpp:18-43).  These pin the oracle; the reference asserts no numbers itself."""
import json
import math
import os

import numpy as np

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def test_clr_pair_order_and_residual():
    k = KAT["clr_J4"]
    x = np.array(k["x"], np.float32)
    P = np.array(k["model"], np.float32)
    r = po.comparer(po.CLR, x, P)
    assert r.tolist() == k["residual"]
    assert float(po.dot_self(r)) == k["distance"]
    assert po.stepper(po.CLR, x, P).tolist() == k["step"]
    assert po.length(po.CLR, 4) == 12 and po.length(po.STANDARD, 9) == 9 and po.length(po.MEDIAN, 7) == 7


def test_neighbourhood_weights():
    for cx, cy, bx, by, sigma, arg in KAT["neighbourhood"]:
        got = po.neighbourhood_weight(cx, cy, bx, by, sigma)
        exp = math.exp(arg) if arg is not None else None
        if sigma > 1.0:
            assert got == exp
    assert po.neighbourhood_weight(3, 3, 3, 3, 1.0) == 1.0
    assert po.neighbourhood_weight(3, 4, 3, 3, 1.0) == 0.0
    assert po.neighbourhood_weight(3, 4, 3, 3, 0.5) == 0.0


def test_batch_accumulator_scalar_chain():
    # 1 node, w == 1: W=1,2,3; M=1,1.5,2; S = 1 + 1 + 2.25 (prefix mean, Q4)
    s = po.OracleSom(1, 1, 1, po.STANDARD)
    X = np.array([[1], [2], [3]], np.float32)
    lb = np.zeros(3, np.uint64)
    mse = s.batch_epoch(X, lb, 5.0, True)
    assert s.map[0, 0] == np.float32(2.0)
    assert s.sigma[0, 0] == np.sqrt(np.float32(4.25) / np.float32(3.0))
    assert s.weight[0] == np.float32(3.0) and s.hits[0] == 3
    # mse over the zero initial map: (1 + 4 + 9)/3 summed in fp32 sample order
    m = np.float32(0)
    for v in (1.0, 4.0, 9.0):
        m = np.float32(m + np.float32(np.float32(v) / np.float32(3)))
    assert mse == m
    s = po.OracleSom(1, 1, 1, po.MEDIAN)
    s.batch_epoch(X, lb, 5.0, True)
    exp = np.float32(1.0)
    exp = np.float32(exp + np.float32(np.float32(0.5) * np.float32(1)))
    exp = np.float32(exp + np.float32(np.float32(np.float32(1) / np.float32(3)) * np.float32(1)))
    assert s.map[0, 0] == exp
    assert s.sigma[0, 0] == np.float32(1.0)       # sqrt(3/3)


def test_local_search_first_candidates_wrap():
    """From node 0 the first candidate set is (W-1,1),(0,1),(1,1),(1,0),(1,H-1),(0,H-1),
    (W-1,H-1),(W-1,0) (Q5): make exactly one of them the best and see it is found."""
    W, H, J = 7, 5, 3
    cands = [(W - 1, 1), (0, 1), (1, 1), (1, 0), (1, H - 1), (0, H - 1), (W - 1, H - 1), (W - 1, 0)]
    v = np.zeros(J, np.float32)
    for cx, cy in cands:
        s = po.OracleSom(W, H, J)
        m = np.full((W * H, J), 5.0, np.float32)
        m[cy * W + cx] = 0.25          # only reachable through wrap-then-clamp
        # block every other route: neighbours of the target stay at 5.0
        s.set_state(map=m)
        got = s.find_local_bmu(v, 0)
        assert got == cy * W + cx, (cx, cy, got)
    # a better node that is NOT in the candidate set is not found
    s = po.OracleSom(W, H, J)
    m = np.full((W * H, J), 5.0, np.float32)
    m[2 * W + 3] = 0.0
    s.set_state(map=m)
    assert s.find_local_bmu(v, 0) == 0


def test_online_window_bounds():
    # b=5, sigma=1 -> i in {2..6}; weights change only at the BMU (sigma<=1) but sigmaMap of the
    # window is rewritten; use sigma=2 at b=0 -> {0..4}
    W = H = 12
    s = po.OracleSom(W, H, 2)
    m = np.full((W * H, 2), 9.0, np.float32)
    m[5 * W + 5] = 0.0
    s.set_state(map=m, S=np.ones((W * H, 2), np.float32))
    s.train_single(np.zeros(2, np.float32), 0.1, 1.0, 5 * W + 5, po.EXPONENTIAL)
    touched = np.argwhere((s.sigma != 0).any(axis=1)).ravel()
    ys, xs = touched // W, touched % W
    assert sorted(set(xs)) == [2, 3, 4, 5, 6] and sorted(set(ys)) == [2, 3, 4, 5, 6]
    s = po.OracleSom(W, H, 2)
    m = np.full((W * H, 2), 9.0, np.float32)
    m[0] = 0.0
    s.set_state(map=m)
    s.train_single(np.zeros(2, np.float32), 0.1, 2.0, 0, po.EXPONENTIAL)
    touched = np.argwhere(s.weight != 0).ravel()
    assert sorted(set(touched % W)) == [0, 1, 2, 3, 4] and sorted(set(touched // W)) == [0, 1, 2, 3, 4]


def test_somindex_divides_by_height():
    s = po.OracleSom(6, 3, 1)
    assert s.somindex(13) == (1, 4)       # (13 - 1) / H(3) = 4, not 13 // W = 2  (Q10)
    s = po.OracleSom(5, 5, 1)
    assert s.somindex(13) == (3, 2)


def test_ican_fixture_shape():
    fx = json.load(open(os.path.join(HERE, "golden", "ican_fixture.json")))
    rows = np.array(fx["rows"], np.float32)
    assert rows.shape == (20, 9)
    assert rows[0].tolist() == [5.0, 0.0, 10.0, 0.0, -1.0, 3.0, 7.0, 3.0, -7.0]
    assert fx["columns"] == list("ABCDEFGHI") and fx["binary"] == [0, 0, 0, 0, 1, 0, 0, 0, 0]
