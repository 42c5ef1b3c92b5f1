"""CPU (build container only): replay the committed goldens on the REAL reference built against real
Eigen (oracle/eigen_crosscheck.cpp, `make -C oracle eigen_crosscheck`) and compare bit for bit.

This is the pin the oracle lacks today ("parity unpinned"): the reference's tests
assert nothing numeric for the hot path and the image has no Eigen, so the recipe stops with "Eigen absent"
and every case below SKIPS with that reason.  On an image that has Eigen (and /root/reference) the same
test turns the goldens -- which the oracle generated and the HIP path is held to -- into reference-pinned
vectors, covering the two conventions the oracle only recalls: Eigen's fp32 reduction order in
dot / squaredNorm (vsom_oracle.c, vso_dot) and sign(NaN).  Nothing here runs on the GPU box."""
import glob
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
GOLD = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ORACLE, "_ref", "eigen_crosscheck")


@pytest.fixture(scope="module")
def crosscheck_exe():
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("reference tree absent (the cross-check is built from /root/reference in the build container only)")
    r = subprocess.run(["make", "-C", ORACLE, "eigen_crosscheck"], capture_output=True, text=True)
    if r.returncode != 0:
        why = (r.stderr.strip().splitlines() or ["build failed"])[0]
        if "Eigen absent" in r.stderr:
            pytest.skip(f"Eigen absent: the reference cannot be built in this image, parity stays unpinned ({why})")
        pytest.fail("eigen_crosscheck recipe failed although Eigen was found:\n" + r.stderr[-3000:])
    return EXE


def _write_case(path, g, mode):
    W, H, J, tr, last = (int(v) for v in g["params"])
    X = np.ascontiguousarray(g["X"], np.float32)
    init = np.ascontiguousarray(g["init_map"], np.float32)
    if mode == 0:
        off = np.asarray(g["chunk_off"], np.int64)
        epochs, fn = last, 0
        par = [float(g["sched"][0]), float(g["sched"][1]), 0.0, 0.0]
    else:
        off = np.array([0, X.shape[0]], np.int64)
        epochs, fn = 1, last
        par = [0.0, 0.0, float(g["sched"][0]), float(g["sched"][1])]
    with open(path, "wb") as f:
        f.write(struct.pack("<8q", W, H, J, tr, mode, epochs, fn, len(off) - 1))
        f.write(struct.pack("<4d", *par))
        f.write(off.tobytes())
        f.write(init.tobytes())
        f.write(X.tobytes())


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


@pytest.mark.parametrize("name", CASES)
def test_reference_with_real_eigen_reproduces_golden(name, crosscheck_exe, tmp_path):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    mode = 0 if "chunk_off" in g.files else 1
    case, dump = str(tmp_path / "case.bin"), str(tmp_path / "dump.bin")
    _write_case(case, g, mode)
    subprocess.run([crosscheck_exe, case, dump], check=True, timeout=600)
    raw = open(dump, "rb").read()
    N, D = g["map"].shape[-2:]
    B = g["X"].shape[0]
    pos = 0

    def take(dtype, count):
        nonlocal pos
        a = np.frombuffer(raw, dtype, count, pos)
        pos += a.nbytes
        return a

    if mode == 0:
        for ep in range(g["map"].shape[0]):
            assert _same(take(np.uint64, B), g["lastbmu"][ep]), (name, ep, "lastBMU")
            assert _same(take(np.float32, N * D).reshape(N, D), g["map"][ep]), (name, ep, "map")
            assert _same(take(np.float32, N * D).reshape(N, D), g["sigma"][ep]), (name, ep, "sigmaMap")
            assert _same(take(np.float32, N), g["weight"][ep]), (name, ep, "weightMap")
            assert _same(take(np.float32, 1)[0], g["mse"][ep]), (name, ep, "MSE")
        assert _same(take(np.uint64, N), g["hits"]), (name, "bmuHits")
    else:
        for k in ("map", "sigma", "S"):
            assert _same(take(np.float32, N * D).reshape(N, D), g[k]), (name, k)
        assert _same(take(np.float32, N), g["weight"]) and _same(take(np.uint64, N), g["hits"])
        assert _same(take(np.uint64, B), g["lastbmu"]) and _same(take(np.float32, 1)[0], g["mse"])
    assert pos == len(raw)
