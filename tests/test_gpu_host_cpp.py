"""GPU: the C++ host mirror (variational-self-organizing-maps_amd/host, libsom_hip.so) driven the
way the reference's perf harness drives libsom; its results are compared bit for bit with the
oracle running the same schedules."""
import math
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "variational-self-organizing-maps_amd", "host")


def make_rows(n, d, seed):
    out = np.empty(n * d, np.float32)
    s = seed
    for i in range(n * d):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = np.float32(np.float32((s >> 8) & 0xFFFF) / np.float32(65536.0) * np.float32(2.0) - np.float32(1.0))
    return out.reshape(n, d)


def read_dump(path):
    raw = open(path, "rb").read()
    N, D, nm = np.frombuffer(raw[:24], np.uint64)
    N, D, nm = int(N), int(D), int(nm)
    off = 24
    out = {}
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
        return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()
    return (a == b).all()


# "single": one GPU.  "group3": VSOM_DEVICES=0,0,0 -- Som then trains through the vsom_group_* entry points
# with three members on the one device (peer-copy transport; sample-sharded phase 1, node-sharded phase 2),
# and every dump must still equal the oracle's
@pytest.fixture(scope="module", params=["single", "group3"])
def dumps(request):
    exe = os.path.join(HOST, "host_api_test")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(HOST, "build.sh")], stdout=subprocess.DEVNULL)
    d = tempfile.mkdtemp(prefix="vsom_host_")
    env = dict(os.environ)
    env.pop("VSOM_DEVICES", None)
    if request.param == "group3":
        env["VSOM_DEVICES"] = "0,0,0"
    res = subprocess.run([exe, d], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert ("group_members=3" if request.param == "group3" else "group_members=1") in res.stdout
    return d, res.stdout, res.stderr


def check_state(dump, o, with_S=False):
    assert beq(dump["map"], o.map) and beq(dump["sigma"], o.sigma)
    assert beq(dump["weight"], o.weight) and beq(dump["hits"], o.hits)
    if with_S:
        assert beq(dump["S"], o.S)


def test_batch_training_through_cpp_api(dumps):
    d, out, _ = dumps
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9, po.STANDARD)
    o.random_initialize(42, 1.0)
    done, mse = o.train_batch(rows, [0, 20, 40, 50], 5, 10.0, 0.3, nthreads=2)
    dump = read_dump(os.path.join(d, "batch_std.bin"))
    check_state(dump, o)
    assert done == 5 and beq(dump["mse"], mse)


def test_batch_training_on_a_preloaded_dataset(dumps):
    """the caller ran loadNextDataFromStream() before train(BatchMap): epoch 0 sees
    hasReadWholeDataStream() already true, trains nothing and records 0/0 = NaN; the stream is reset and
    epochs 1.. train with the local search (Som.cpp:735-749; isFirst only for i == 0, :738)"""
    d, out, err = dumps
    assert "no prefetched chunk" not in out + err
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9, po.STANDARD)
    o.random_initialize(42, 1.0)
    mse = [np.float32(np.nan)]
    for e in (1, 2):
        lb = np.zeros(50, np.uint64)                     # lastBMU zeroed by the reload
        mse.append(np.float32(o.batch_epoch(rows, lb, 6.0 * math.exp(-0.2 * e), False)))
    dump = read_dump(os.path.join(d, "batch_preloaded.bin"))
    check_state(dump, o)
    assert beq(dump["mse"], np.array(mse, np.float32))


def test_online_then_batch_on_the_same_map(dumps):
    """two online epochs, then two batch-map epochs (with a group: member 0 trained alone, replicas re-synchronised)"""
    d, _, _ = dumps
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9, po.STANDARD)
    o.random_initialize(13, 1.0)
    mse1 = o.train_online(rows, [0, 20, 40, 50], 2, 0.05, 0.1, 4.0, 0.3, po.EXPONENTIAL)
    done, mse2 = o.train_batch(rows, [0, 20, 40, 50], 2, 5.0, 0.2, nthreads=2)
    assert done == 2
    dump = read_dump(os.path.join(d, "online_then_batch.bin"))
    check_state(dump, o, with_S=True)
    assert beq(dump["mse"], np.concatenate([mse1, mse2]).astype(np.float32))


def test_online_training_through_cpp_api(dumps):
    d, _, _ = dumps
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9, po.MEDIAN)
    o.random_initialize(7, 1.0)
    mse = o.train_online(rows, [0, 20, 40, 50], 3, 0.05, 0.1, 3.0, 0.5, po.EXPONENTIAL)
    dump = read_dump(os.path.join(d, "online_median.bin"))
    check_state(dump, o, with_S=True)
    assert beq(dump["mse"], mse)

    crows = make_rows(30, 5, 777)
    o = po.OracleSom(6, 6, 5, po.CLR)
    o.random_initialize(3, 1.0)
    mse = o.train_online(crows, [0, 30], 2, 0.01, 0.0, 2.0, 0.2, po.INVERSE_PROPORTIONAL)
    dump = read_dump(os.path.join(d, "online_clr.bin"))
    check_state(dump, o, with_S=True)
    assert beq(dump["mse"], mse)


def test_online_schedule_down_to_local_walks_through_cpp_api(dumps):
    """twelve online epochs on the reference's scenario shape (10 x 10 x 9, chunks of 20 / 20 / 10): sigma 3 -> 1.1 over five
    epochs, then clamped at 1 (Som.cpp:1148-1149) where trainSingle walks from lastBMU (:891) -- every chunk of every epoch
    one launch (online_tiny_chunk_kernel), the running MSE carried across the chunks of an epoch"""
    d, _, _ = dumps
    rows = make_rows(50, 9, 12345)
    for name, tr, fn in (("online_to_local_std.bin", po.STANDARD, po.INVERSE_PROPORTIONAL),
                         ("online_to_local_median.bin", po.MEDIAN, po.EXPONENTIAL)):
        o = po.OracleSom(10, 10, 9, tr)
        o.random_initialize(11, 1.0)
        mse = o.train_online(rows, [0, 20, 40, 50], 12, 0.05, 0.1, 3.0, 0.25, fn)
        dump = read_dump(os.path.join(d, name))
        check_state(dump, o, with_S=True)
        assert beq(dump["mse"], mse), name


def test_search_single_copy_and_checkpoint(dumps):
    d, _, _ = dumps
    rows = make_rows(50, 9, 12345)
    v = rows[0]
    o = po.OracleSom(10, 10, 9)
    o.random_initialize(11, 1.0)
    b = o.find_bmu(v)
    loc = o.find_local_bmu(v, 37)
    dist = o.dist(b, v)
    bmu, res, derr, last = o.train_single(v, 0.1, 2.0, 5, po.EXPONENTIAL)
    tok = open(os.path.join(d, "search.txt")).read().split()
    assert int(tok[0]) == b and int(tok[1]) == loc and float.fromhex(tok[2]) == dist
    assert int(tok[3]) == bmu and int(tok[4]) == last
    assert np.float32(float.fromhex(tok[5])) == derr and np.float32(float.fromhex(tok[6])) == res[0]
    for name in ("single.bin", "single_copy.bin", "single_loaded.bin"):
        check_state(read_dump(os.path.join(d, name)), o, with_S=True)


def test_octave_text_checkpoint(dumps):
    """Som::save writes the reference's text format (Som.cpp:1209-1294) byte for byte; Som(const
    char*) + Som::load read it back (six decimals, no SMap -- as the reference)."""
    import octave_text
    d, _, _ = dumps
    src = read_dump(os.path.join(d, "single_text_src.bin"))
    U = np.array([float.fromhex(t) for t in open(os.path.join(d, "umatrix.txt")).read().split()])
    want = octave_text.render(10, 10, 9, src["map"], src["sigma"], src["weight"], src["hits"], U)
    assert src["hits"][4 * 10 + 3] == 2                     # the two addBmu(SomIndex(3,4)) calls
    assert open(os.path.join(d, "ckpt.txt")).read() == want
    assert open(os.path.join(d, "ckpt2.txt")).read() == want
    got = read_dump(os.path.join(d, "single_text_loaded.bin"))
    assert (got["map"] == octave_text.round6(src["map"]).astype(np.float32)).all()
    assert (got["sigma"] == octave_text.round6(src["sigma"]).astype(np.float32)).all()
    assert (got["weight"] == octave_text.round6(src["weight"]).astype(np.float32)).all()
    assert (got["hits"] == src["hits"]).all()
    assert (got["S"] == 0).all()                            # not part of the format (Som.cpp:51-83 zeroes it)


def test_mnist_loader_to_batch_training_end_to_end(tmp_path):
    """BASELINE configuration 2's plumbing (IDX -> MnistDataLoader -> DataSet -> Som::train BatchMap)
    against the oracle on the same rows and chunk boundaries.  A chunked MnistDataLoader pass ends
    with a zero-row load (MnistDataLoader.cpp:49-55) and the reference's epoch over that empty chunk
    rewrites every neuron (zero vector, NaN sigma, zero weight, Som.cpp:840-875) -- reproduced."""
    import struct
    rs = np.random.RandomState(11)
    n = 600
    images = (rs.randint(0, 256, size=(n, 28, 28)) * (rs.rand(n, 28, 28) < 0.2)).astype(np.uint8)
    labels = rs.randint(0, 10, size=n).astype(np.uint8)
    folder = str(tmp_path)
    with open(os.path.join(folder, "train-images-idx3-ubyte"), "wb") as f:
        f.write(struct.pack(">IIII", 0x803, n, 28, 28) + images.tobytes())
    with open(os.path.join(folder, "train-labels-idx1-ubyte"), "wb") as f:
        f.write(struct.pack(">II", 0x801, n) + labels.tobytes())
    exe = os.path.join(HOST, "host_api_test")
    res = subprocess.run([exe, "mnist", folder, folder], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-400:] + res.stderr[-400:]
    rows = np.concatenate([images.reshape(n, 784), np.eye(10)[labels]], axis=1).astype(np.float32)
    o = po.OracleSom(12, 12, 794, po.STANDARD)
    o.random_initialize(5, 1.0)
    done, mse = o.train_batch(rows, [0, 256, 512, 600, 600], 3, 6.0, 0.2, nthreads=4)   # 256, 256, 88, 0 rows
    dump = read_dump(os.path.join(folder, "mnist_batch.bin"))
    check_state(dump, o)
    assert done == 3 and beq(dump["mse"], mse)
    assert (dump["map"] == 0).all() and np.isnan(dump["sigma"]).all()    # the empty chunk's doing


def test_custom_transformation_trains_on_the_host_without_a_device_context(dumps):
    """caller-supplied hooks: no device context, training on the host (tests/test_host_custom.py holds that
    path and its consumers -- U-matrix, evaluate, ... -- to the oracle)"""
    _, out, err = dumps
    assert "custom_transformation_host_path=1 consumers_run=1 kind=-1" in out, out + err


def test_next_rows_restricted_bmu_bmd_umatrix_evaluate(dumps):
    d, _, _ = dumps
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9)
    o.random_initialize(21, 1.0)
    o.train_batch(rows, [0, 50], 2, 4.0, 0.2, nthreads=2)
    st = read_dump(os.path.join(d, "next_state.bin"))
    check_state(st, o)
    v = rows[3]
    lines = open(os.path.join(d, "next_rows.txt")).read().split("\n")
    a, b, c = [int(t) for t in lines[0].split()]
    assert (a, b, c) == (o.find_restricted_bmu(v, 1), o.find_restricted_bmu(v, 3), o.find_restricted_bmu(v, 1000))
    assert c == 0                                          # nothing qualifies: node 0 seeds (Som.cpp:316-317)
    bmd = np.fromfile(os.path.join(d, "bmd.bin"), np.float64)
    exp = o.find_restricted_bmd(v, 1)
    assert (bmd == exp).all() or np.allclose(bmd, exp, rtol=1e-15, atol=0)
    um = np.fromfile(os.path.join(d, "umatrix.bin"), np.float64)
    assert (um == o.update_umatrix()).all()
    tok = lines[1].split()
    assert float.fromhex(tok[0]) == o.dist_raw(17, v)
    # evaluate on all-continuous data = running mean of the BMU distances (Som.cpp:519)
    err = 0.0
    for i in range(50):
        err += 1.0 / (i + 1.0) * (o.dist(o.find_bmu(rows[i]), rows[i]) - err)
    assert float.fromhex(tok[1]) == err
    # measureSimilarity (Som.cpp:631-714) restated
    def measure(nsig, minhits):
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
    # CLR U-matrix
    oc = po.OracleSom(5, 4, 4, po.CLR)
    oc.random_initialize(5, 1.0)
    oc.sigma[...] = (0.25 + 0.01 * (np.arange(20 * 12) % 7)).astype(np.float32).reshape(20, 12)
    umc = np.fromfile(os.path.join(d, "umatrix_clr.bin"), np.float64)
    assert (umc == oc.update_umatrix()).all()


def test_umatrix_after_every_epoch_matches_the_final_state(dumps):
    """train(..., BatchMap, updateUMatrixAfterEpoch = true) (Som.cpp:751-752): the U-matrix left behind is that of
    the last epoch's map and sigmaMap -- also under a group, whose deferred sigmaMap gather updateUMatrix has
    to join first (r2 advisor finding)"""
    d, _, _ = dumps
    rows = make_rows(50, 9, 12345)
    o = po.OracleSom(10, 10, 9)
    o.random_initialize(33, 1.0)
    o.train_batch(rows, [0, 20, 40, 50], 3, 5.0, 0.2, nthreads=2)
    check_state(read_dump(os.path.join(d, "umatrix_after_epoch_state.bin")), o)
    um = np.fromfile(os.path.join(d, "umatrix_after_epoch.bin"), np.float64)
    assert (um == o.update_umatrix()).all()
