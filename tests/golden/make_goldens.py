#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ with the CPU oracle.

The reference cannot be built or imported here (C++ needing Eigen/doctest/sqlite3.c, all absent),
so these vectors are ORACLE outputs (parity unpinned, see oracle/vsom_oracle.h): they freeze the
oracle's behaviour (regression) and give the HIP path fixed expected outputs that do not depend on
the oracle being rebuilt.  Inputs are seeded (tests/gen.py) or the reference's own 20-row SQLite
fixture (ican_fixture.json, extracted with `--fixture` where /root/reference exists).

  python tests/golden/make_goldens.py            # rewrite *.npz
  python tests/golden/make_goldens.py --fixture  # also re-extract ican_fixture.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import gen  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def extract_fixture():
    import sqlite3
    c = sqlite3.connect("file:/root/reference/tests/performance/data/testDb.sq3?mode=ro", uri=True)
    rows = [list(r[1:]) for r in c.execute("select Id,A,B,C,D,E,F,G,H,I from Ican order by Id")]
    json.dump({"_doc": "Rows of table Ican in the reference's tests/performance/data/testDb.sq3 "
                       "(data fixture) and its column spec (columnSpec.txt, perf_tests.cpp:35-58)",
               "columns": list("ABCDEFGHI"), "binary": [0, 0, 0, 0, 1, 0, 0, 0, 0],
               "weights": [1] * 9, "rows": rows}, open(os.path.join(HERE, "ican_fixture.json"), "w"), indent=1)


def batch_case(name, W, H, J, tr, X, init, sigma0, decay, epochs, chunk):
    """trainBatchSom over chunks of `chunk` rows; records per-epoch state."""
    B = X.shape[0]
    off = list(range(0, B, chunk)) + [B]
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    out = {"X": X, "init_map": init, "chunk_off": np.array(off, np.int64),
           "params": np.array([W, H, J, tr, epochs], np.int64), "sched": np.array([sigma0, decay])}
    maps, sigmas, weights, lbs, mses = [], [], [], [], []
    for ep in range(epochs):
        sigma = sigma0 * np.exp(-decay * ep)
        if sigma < 1.0:
            break
        mse = np.float32(0)
        lb_all = np.zeros(B, np.uint64)
        for c in range(len(off) - 1):
            lb = np.zeros(off[c + 1] - off[c], np.uint64)
            mse = np.float32(mse + o.batch_epoch(X[off[c]:off[c + 1]], lb, sigma, ep == 0))
            lb_all[off[c]:off[c + 1]] = lb
        mses.append(np.float32(mse / np.float32(len(off) - 1)))
        maps.append(o.map.copy()); sigmas.append(o.sigma.copy()); weights.append(o.weight.copy())
        lbs.append(lb_all)
    out.update(map=np.stack(maps), sigma=np.stack(sigmas), weight=np.stack(weights),
               lastbmu=np.stack(lbs), mse=np.array(mses, np.float32), hits=o.hits.copy())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "epochs", len(maps), "nan", bool(np.isnan(maps[-1]).any()))


def online_case(name, W, H, J, tr, X, init, eta, sigma, fn):
    B = X.shape[0]
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    mse = o.train_online_chunk(X, lb, eta, sigma, fn)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), X=X, init_map=init,
                        params=np.array([W, H, J, tr, fn], np.int64), sched=np.array([eta, sigma]),
                        map=o.map.copy(), sigma=o.sigma.copy(), S=o.S.copy(), weight=o.weight.copy(),
                        hits=o.hits.copy(), lastbmu=lb, mse=np.float32(mse))
    print(name, "mse", float(mse))


def main():
    if "--fixture" in sys.argv:
        extract_fixture()
    fx = np.array(json.load(open(os.path.join(HERE, "ican_fixture.json")))["rows"], np.float32)
    # the reference's own perf scenario: 10x10 map on the 20-row fixture (perf_tests.cpp:74-112)
    batch_case("fixture_batch", 10, 10, 9, po.STANDARD, fx, gen.random_map(100, 9, 42), 10.0, 0.01, 4, 20)
    online_case("fixture_online_exp", 10, 10, 9, po.STANDARD, fx, gen.random_map(100, 9, 42), 0.001, 10.0, po.EXPONENTIAL)
    online_case("fixture_online_inv", 10, 10, 9, po.STANDARD, fx, gen.random_map(100, 9, 42), 0.001, 10.0, po.INVERSE_PROPORTIONAL)
    # C1: 10x10 map, 16-dim blobs, two chunks
    batch_case("c1_batch", 10, 10, 16, po.STANDARD, gen.blobs(96, 16, 4, 1, 2), gen.random_map(100, 16, 42), 5.0, 0.05, 3, 64)
    online_case("c1_online_exp", 10, 10, 16, po.STANDARD, gen.blobs(48, 16, 4, 1, 2), gen.random_map(100, 16, 42), 0.1, 3.0, po.EXPONENTIAL)
    # C4 (reduced): Median estimator
    batch_case("c4_median", 12, 12, 32, po.MEDIAN, gen.blobs(80, 32, 8, 4, 4, sigma=0.5), gen.random_map(144, 32, 4), 4.0, 0.1, 2, 80)
    online_case("c4_median_online_inv", 8, 8, 12, po.MEDIAN, gen.blobs(40, 12, 4, 4, 4, sigma=0.5), gen.random_map(64, 12, 4), 0.1, 2.0, po.INVERSE_PROPORTIONAL)
    # C5 (reduced): combinatorial linear regression, J=6 -> D=30
    batch_case("c5_clr", 6, 6, 6, po.CLR, gen.correlated(48, 6, 5), gen.random_map(36, 30, 5), 3.0, 0.1, 2, 48)
    online_case("c5_clr_online_exp", 6, 6, 5, po.CLR, gen.correlated(24, 5, 5), gen.random_map(36, 20, 5), 0.01, 2.0, po.EXPONENTIAL)
    # local search + sigma <= 1 online
    online_case("online_local_sigma1", 9, 9, 7, po.STANDARD, gen.blobs(30, 7, 3, 1, 2), gen.random_map(81, 7, 6), 0.05, 1.0, po.EXPONENTIAL)
    # ---- the quirk list, so that a replay on the real reference (oracle/eigen_crosscheck.cpp)
    #      pins every convention the oracle only restates ----
    # Q10: SomIndex divides by HEIGHT (SomIndex.cpp:15-18) -- wrong for W != H; both orientations, local epochs incl.
    batch_case("q10_nonsquare_9x7_batch", 9, 7, 13, po.STANDARD, gen.blobs(70, 13, 4, 1, 2, sigma=0.3), gen.random_map(63, 13, 8), 4.0, 0.2, 3, 40)
    batch_case("q10_nonsquare_5x11_median_batch", 5, 11, 10, po.MEDIAN, gen.blobs(60, 10, 4, 3, 5, sigma=0.4), gen.random_map(55, 10, 9), 3.0, 0.15, 3, 60)
    online_case("q10_nonsquare_7x11_online_exp", 7, 11, 9, po.STANDARD, gen.blobs(36, 9, 3, 1, 2, sigma=0.3), gen.random_map(77, 9, 10), 0.1, 2.5, po.EXPONENTIAL)
    # Q7: weight underflow -> 0/0 poisons nodes (Som.cpp:857-864): 36x36 map, sigma 2 and 1.64: nodes farther than
    # ~29 cells from the first sample's BMU start from W = 0 and stay NaN for the epoch
    batch_case("q7_nan_poison_36x36_batch", 36, 36, 6, po.STANDARD, gen.blobs(40, 6, 3, 1, 2, sigma=0.3), gen.random_map(1296, 6, 11), 2.0, 0.2, 2, 40)
    # Q1: depth 794 = what MnistDataLoader yields (784 pixels + 10 one-hot, MnistDataLoader.cpp:47-84): 99 packets
    # of 8 and a two-element tail in Eigen's dot (Som.cpp:140)
    batch_case("q1_depth794_tail_batch", 6, 5, 794, po.STANDARD, gen.mnist_like(24, 3, 794), gen.random_map(30, 794, 12) * np.float32(100), 2.5, 0.3, 2, 24)
    # D = 9 + D = 13 are covered above (one packet + tail); D = 3 (< one packet): sequential sum
    batch_case("q1_depth3_batch", 8, 8, 3, po.STANDARD, gen.blobs(30, 3, 3, 1, 2, sigma=0.3), gen.random_map(64, 3, 13), 3.0, 0.3, 2, 30)
    # CLR with P = J(J-1)/2 not a multiple of 4 (J = 7: P = 21, one whole packet pair + 5) on a non-square map
    batch_case("q1_clr_p21_5x4_batch", 5, 4, 7, po.CLR, gen.correlated(40, 7, 5), gen.random_map(20, 42, 14), 2.5, 0.2, 2, 40)
    online_case("q1_clr_p21_online_inv", 5, 4, 7, po.CLR, gen.correlated(20, 7, 5), gen.random_map(20, 42, 14), 0.01, 2.0, po.INVERSE_PROPORTIONAL)
    # InverseProportional from the all-zero weight map Construct leaves (Som.cpp:928-936: t = Wt == 0 ? 1 : h/Wt)
    # on the reference's 20-row fixture is fixture_online_inv above; the same with sigma <= 1 (local search, window
    # of one node, most weights stay 0):
    online_case("fixture_online_inv_sigma1", 10, 10, 9, po.STANDARD, fx, gen.random_map(100, 9, 42), 0.001, 1.0, po.INVERSE_PROPORTIONAL)


if __name__ == "__main__":
    main()
