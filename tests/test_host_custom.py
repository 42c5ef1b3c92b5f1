"""CPU: the C++ mirror's host path for caller-supplied Transformation hooks (host/src/vsom_custom.cpp).

The reference lets a caller build a Transformation from own lambdas (include/Transformation.hpp:13-28,
tests/test1.cpp:46-90 "Fakes"); such hooks cannot run on the GPU, so the mirror keeps that Som's state on
the host and runs distance / findBmu / findLocalBmu / batch epoch / online step there (Som.cpp:124-141,
291-454, 716-947, 1135-1187).  host_custom_test trains with lambdas that compute what Standard and
StandardMedianEstimator compute; the results must equal the oracle's for those transformations BIT FOR
BIT (the oracle here is the checker; the product code under test is the mirror's own C++).  No GPU."""
import math
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "variational-self-organizing-maps_amd", "host")
EXE = os.path.join(HOST, "host_custom_test")


def make_rows(n, d, seed):
    out = np.empty(n * d, np.float32)
    s = seed
    for i in range(n * d):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = np.float32(np.float32((s >> 8) & 0xFFFF) / np.float32(65536.0) * np.float32(2.0) - np.float32(1.0))
    return out.reshape(n, d)


def read_dump(path):
    raw = open(path, "rb").read()
    N, D, nm = (int(v) for v in np.frombuffer(raw[:24], np.uint64))
    off, out = 24, {}
    for k in ("map", "sigma", "S"):
        out[k] = np.frombuffer(raw, np.float32, N * D, off).reshape(N, D)
        off += N * D * 4
    out["weight"] = np.frombuffer(raw, np.float32, N, off)
    off += N * 4
    out["hits"] = np.frombuffer(raw, np.uint64, N, off)
    off += N * 8
    out["mse"] = np.frombuffer(raw, np.float32, nm, off)
    return out


def beq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


@pytest.fixture(scope="module")
def outdir():
    if not os.path.exists(EXE):
        import __graft_entry__
        __graft_entry__.build()
    d = tempfile.mkdtemp(prefix="vsom_custom_")
    r = subprocess.run([EXE, d], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host_custom_test done" in r.stdout, r.stdout + r.stderr
    return d


def check(dump, o, mse, with_S=False):
    assert beq(dump["map"], o.map) and beq(dump["sigma"], o.sigma)
    assert beq(dump["weight"], o.weight) and beq(dump["hits"], o.hits)
    if with_S:
        assert beq(dump["S"], o.S)
    assert beq(dump["mse"], np.asarray(mse, np.float32))


def test_fakes_transformation_of_the_reference_tests(outdir):
    """tests/test1.cpp:46-90: P = 1..8, X = 1..4, Comparer = Stepper = A.*X + B  ->  6 10 16 24"""
    lines = open(os.path.join(outdir, "fakes.txt")).read().splitlines()
    assert lines[0] == "-1"                                   # kind(): Custom
    assert lines[1].split() == ["6", "10", "16", "24"] and lines[2].split() == ["6", "10", "16", "24"]
    assert lines[3] == "4 Standard transformation"            # default Length / Name members (Transformation.hpp:31-39)
    assert lines[4] == "8 3 2 1"                              # depth-8 map of 3x2, no device context


def test_batch_training_with_standard_equivalent_hooks(outdir):
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9, po.STANDARD)
    o.random_initialize(42, 1.0)
    done, mse = o.train_batch(rows, [0, 20, 40, 50], 5, 10.0, 0.3, nthreads=2)
    assert done == 5
    check(read_dump(os.path.join(outdir, "custom_batch_std.bin")), o, mse)
    # searches on the trained map: findBmu, findLocalBmu from node 37, the distance to the BMU
    v = rows[7]
    bmu = o.find_bmu(v)
    loc = o.find_local_bmu(v, 37)
    dist = o.dist(bmu, v)
    got = open(os.path.join(outdir, "custom_search.txt")).read().split()
    assert int(got[0]) == bmu and int(got[1]) == loc
    assert np.float32(float.fromhex(got[2])) == np.float32(dist)


def test_consumers_outside_training_with_custom_hooks(outdir):
    """findRestrictedBmu / findRestrictedBmd / euclidianWeightedDistRaw / updateUMatrix / evaluate /
    measureSimilarity of a custom-hook Som (Som.cpp:143-157, 313-332, 457-523, 631-714, 999-1111) against the
    oracle on the map test_batch_training_with_standard_equivalent_hooks trains"""
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9, po.STANDARD)
    o.random_initialize(42, 1.0)
    done, mse = o.train_batch(rows, [0, 20, 40, 50], 5, 10.0, 0.3, nthreads=2)
    v = rows[7]
    lines = open(os.path.join(outdir, "custom_consumers.txt")).read().split("\n")
    a, b, c = [int(t) for t in lines[0].split()]
    assert (a, b, c) == (o.find_restricted_bmu(v, 1), o.find_restricted_bmu(v, 2), o.find_restricted_bmu(v, 1000))
    assert c == 0                                          # nothing qualifies: node 0 seeds (Som.cpp:316-317)
    bmd = np.fromfile(os.path.join(outdir, "custom_bmd.bin"), np.float64)
    exp = o.find_restricted_bmd(v, 1)
    assert (bmd == exp).all() or np.allclose(bmd, exp, rtol=1e-15, atol=0)
    um = np.fromfile(os.path.join(outdir, "custom_umatrix.bin"), np.float64)
    assert (um == o.update_umatrix()).all()
    tok = lines[1].split()
    assert float.fromhex(tok[0]) == o.dist_raw(17, v)
    err = 0.0                                              # evaluate on all-continuous data (Som.cpp:519)
    for i in range(50):
        err += 1.0 / (i + 1.0) * (o.dist(o.find_bmu(rows[i]), rows[i]) - err)
    assert float.fromhex(tok[1]) == err

    def measure(nsig, minhits):                            # Som.cpp:631-714 restated
        maxv, maxrow, last, success = np.float32(-99999999.0), 0, False, True
        i = 0
        while i < 51:
            if i == 50:
                i, last = maxrow, True
            pos = o.find_restricted_bmu(rows[i], minhits)
            sg, m = o.sigma[pos], o.map[pos]
            sM = np.where(sg > np.float32(1e-5), np.float32(1e-5), sg).astype(np.float32)
            with np.errstate(all="ignore"):
                delta = ((rows[i] - m) / sM / np.float32(nsig)).astype(np.float32)
            mn, mx = m - sM * np.float32(nsig), m + sM * np.float32(nsig)
            for k in range(9):
                if delta[k] > maxv:
                    maxv, maxrow = np.float32(abs(delta[k])), i
                if last and (rows[i][k] < mn[k] or rows[i][k] > mx[k]):
                    success = False
            if last:
                break
            i += 1
        return int(success)
    assert int(tok[2]) == measure(3, 1) and int(tok[3]) == measure(1000000, 1)
    assert lines[2].strip() == "1"
    # copies of a trained custom-hook Som carry its state (they used to come out zeroed)
    for name in ("custom_copy.bin", "custom_assigned.bin"):
        check(read_dump(os.path.join(outdir, name)), o, mse)
    # Octave text checkpoint round trip (six decimals, Som.cpp:1209-1294)
    m52, w5 = (float(t) for t in lines[3].split())
    assert abs(m52 - float(o.map[5, 2])) < 1e-5 and abs(w5 - float(o.weight[5])) < 1e-4 * max(1.0, abs(float(o.weight[5])))


def test_batch_training_with_median_equivalent_hooks(outdir):
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9, po.MEDIAN)
    o.random_initialize(9, 1.0)
    done, mse = o.train_batch(rows, [0, 20, 40, 50], 3, 6.0, 0.2, nthreads=2)
    assert done == 3
    check(read_dump(os.path.join(outdir, "custom_batch_median.bin")), o, mse)


def test_online_training_with_custom_hooks(outdir):
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9, po.MEDIAN)
    o.random_initialize(7, 1.0)
    mse = o.train_online(rows, [0, 20, 40, 50], 3, 0.05, 0.1, 3.0, 0.5, po.EXPONENTIAL)
    check(read_dump(os.path.join(outdir, "custom_online_median.bin")), o, mse, with_S=True)
    o = po.OracleSom(10, 10, 9, po.STANDARD)
    o.random_initialize(3, 1.0)
    mse = o.train_online(rows, [0, 50], 3, 0.01, 0.0, 2.0, 0.7, po.INVERSE_PROPORTIONAL)   # sigma: 2, 1 (clamped), 1
    check(read_dump(os.path.join(outdir, "custom_online_inv.bin")), o, mse, with_S=True)
    assert math.isfinite(float(mse[-1]))
