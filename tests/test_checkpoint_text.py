"""The reference's Octave/Matlab text checkpoint in the host library
(vsom_checkpoint.cpp): a file written in the reference's format (restated in tests/octave_text.py
from src/Som.cpp:1209-1294) is sized (getSizeFromFile), loaded (Som::load) and written back
byte-identically.  No GPU needed."""
import os
import subprocess

import numpy as np
import pytest

import octave_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "variational-self-organizing-maps_amd", "host", "host_loader_test")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        import __graft_entry__
        __graft_entry__.build()
    return EXE


@pytest.mark.parametrize("W,H,D", [(10, 10, 9), (4, 4, 1), (16, 16, 33)])
def test_reference_format_round_trip(exe, tmp_path, W, H, D):
    rs = np.random.RandomState(W * 100 + D)
    N = W * H
    m = (rs.randn(N, D) * 50).astype(np.float32)
    m[0, 0] = 0.0
    m[1, 0] = -0.0000004          # prints as -0.000000
    m[2, 0] = 123456.789
    s = np.abs(rs.randn(N, D)).astype(np.float32)
    w = np.abs(rs.randn(N) * 10).astype(np.float32)
    hits = rs.randint(0, 5000, size=N).astype(np.uint64)
    U = np.abs(rs.randn(N))
    text = octave_text.render(W, H, D, m, s, w, hits, U)
    src, dst = os.path.join(str(tmp_path), "in.txt"), os.path.join(str(tmp_path), "out.txt")
    open(src, "w").write(text)
    r = subprocess.run([exe, "octave", src, dst], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["DIMS", str(W), str(H), str(D)]
    assert open(dst).read() == text


def test_non_square_dims_follow_the_reference(exe, tmp_path):
    """getSizeFromFile takes height from '# rows' and width from '# columns' (Som.cpp:1322-1333)
    while save writes rows = width, columns = height (:1255-1256): a 6x3 map is sized 3x6."""
    W, H, D = 6, 3, 2
    N = W * H
    z = np.zeros((N, D), np.float32)
    text = octave_text.render(W, H, D, z, z, np.zeros(N), np.zeros(N), np.zeros(N))
    src, dst = os.path.join(str(tmp_path), "in.txt"), os.path.join(str(tmp_path), "out.txt")
    open(src, "w").write(text)
    r = subprocess.run([exe, "octave", src, dst], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["DIMS", "3", "6", "2"]
