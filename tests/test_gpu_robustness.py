"""GPU: the drop-in boundary under concurrency and exhaustion (SURVEY 8b's threading contract, include/SOM.hpp:63,76).

  * two contexts on one device driven from two host threads, each through a five-epoch schedule: both bit-exact;
  * the C++ mirror's Som::train in a worker thread while another polls isTraining() / getMetrics() under metricsMutex,
    and two Som objects training at once from two threads (host_api_test threads);
  * 200 create / train-one-chunk / destroy cycles leave the device's free memory where it was;
  * an allocation that cannot fit returns VSOM_ERR_NOMEM and the same context then trains a chunk bit-exactly.
Everything is compared with the oracle running the same schedules alone."""
import ctypes as C
import os
import subprocess
import tempfile
import threading

import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po
from test_gpu_host_cpp import make_rows, read_dump, beq, check_state, HOST

pytestmark = pytest.mark.gpu


def _schedule_on_context(W, H, J, tr, chunks, init, sigmas, out, device=0):
    """five batch epochs through the C ABI on a context of its own; everything the schedule produces goes to `out`"""
    try:
        ctx = vsom_amd.Context(W, H, J, tr, device=device)
        ctx.set_state(map=init)
        res = []
        for e, sigma in enumerate(sigmas):
            x = chunks[e % len(chunks)]
            ctx.upload_chunk(x)
            mse = ctx.batch_epoch(sigma, e < 2)
            res.append((np.float32(mse), ctx.get_last_bmu().copy()))
        out["state"] = ctx.get_state()
        out["epochs"] = res
        ctx.close()
    except Exception as ex:      # surfaced by the asserting thread below
        out["error"] = repr(ex)


def _schedule_on_oracle(W, H, J, tr, chunks, init, sigmas):
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    res = []
    for e, sigma in enumerate(sigmas):
        x = chunks[e % len(chunks)]
        lb = np.zeros(x.shape[0], np.uint64)
        mse = o.batch_epoch(x, lb, sigma, e < 2, nthreads=8)
        res.append((np.float32(mse), lb))
    return o, res


def test_two_contexts_from_two_host_threads():
    """vsom_last_error is thread-local, every entry point selects its context's device, and two contexts share nothing
    but the device: two threads driving one context each must both reproduce the oracle."""
    jobs = [
        dict(W=40, H=36, J=96, tr=po.STANDARD, seeds=(3, 4, 5), B=1300, scale=100.0),
        dict(W=32, H=32, J=40, tr=po.MEDIAN, seeds=(6, 7), B=900, scale=1.0),
    ]
    sigmas = [9.0, 7.5, 6.0, 5.0, 4.0]
    work = []
    for j in jobs:
        if j["tr"] == po.STANDARD:
            chunks = [gen.mnist_like(j["B"], seed=s, dim=j["J"]) for s in j["seeds"]]
        else:
            chunks = [gen.blobs(j["B"], j["J"], 5, 1, s, sigma=0.4) for s in j["seeds"]]
        init = (gen.random_map(j["W"] * j["H"], j["J"], seed=42) * np.float32(j["scale"])).astype(np.float32)
        work.append((j, chunks, init))
    outs = [dict(), dict()]
    threads = [threading.Thread(target=_schedule_on_context,
                                args=(j["W"], j["H"], j["J"], j["tr"], chunks, init, sigmas, outs[i]))
               for i, (j, chunks, init) in enumerate(work)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
        assert not t.is_alive()
    for i, (j, chunks, init) in enumerate(work):
        assert "error" not in outs[i], outs[i].get("error")
        o, want = _schedule_on_oracle(j["W"], j["H"], j["J"], j["tr"], chunks, init, sigmas)
        for e, ((mse_g, lb_g), (mse_o, lb_o)) in enumerate(zip(outs[i]["epochs"], want)):
            assert beq(lb_g, lb_o), (i, e, "lastBMU")
            assert beq(mse_g, mse_o), (i, e, mse_g, mse_o)
        st = outs[i]["state"]
        for k in ("map", "sigma", "weight", "hits"):
            assert beq(st[k], getattr(o, k)), (i, k)


@pytest.fixture(scope="module")
def thread_dumps():
    exe = os.path.join(HOST, "host_api_test")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(HOST, "build.sh")], stdout=subprocess.DEVNULL)
    d = tempfile.mkdtemp(prefix="vsom_thr_")
    env = dict(os.environ)
    env.pop("VSOM_DEVICES", None)
    res = subprocess.run([exe, "threads", d], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    return d, res.stdout, res.stderr


def test_polling_thread_beside_train(thread_dumps):
    """Som::train(BatchMap) in a worker thread; the main thread polls isTraining() and copies getMetrics() under
    metricsMutex until the worker ends (SOM.hpp:63,76).  The poller saw the flag set, never saw a half-built metrics vector,
    the flag is clear afterwards, and the trained map is the oracle's."""
    d, out, err = thread_dumps
    line = [ln for ln in out.splitlines() if ln.startswith("poll:")][0]
    kv = dict(tok.split("=") for tok in line.split()[1:])
    assert int(kv["polls"]) > 0 and kv["seen_training"] == "1" and kv["sizes_ok"] == "1" and kv["still_training"] == "0", line
    rows = make_rows(600, 16, 4242)
    o = po.OracleSom(24, 20, 16, po.STANDARD)
    o.random_initialize(42, 1.0)
    done, mse = o.train_batch(rows, [0, 200, 400, 600], 6, 8.0, 0.2, nthreads=4)
    dump = read_dump(os.path.join(d, "poll_batch.bin"))
    check_state(dump, o)
    assert done == 6 and beq(dump["mse"], mse)


def test_two_soms_training_at_once(thread_dumps):
    d, out, err = thread_dumps
    rows_a, rows_b = make_rows(600, 16, 4242), make_rows(600, 16, 777)
    a = po.OracleSom(24, 20, 16, po.STANDARD)
    a.random_initialize(5, 1.0)
    done, mse_a = a.train_batch(rows_a, [0, 200, 400, 600], 5, 7.0, 0.25, nthreads=4)
    dump = read_dump(os.path.join(d, "thr_a.bin"))
    check_state(dump, a)
    assert done == 5 and beq(dump["mse"], mse_a)
    b = po.OracleSom(24, 20, 16, po.MEDIAN)
    b.random_initialize(6, 1.0)
    mse_b = b.train_online(rows_b, [0, 200, 400, 600], 3, 0.05, 0.1, 3.0, 0.3, po.EXPONENTIAL)
    dump = read_dump(os.path.join(d, "thr_b.bin"))
    check_state(dump, b, with_S=True)
    assert beq(dump["mse"], mse_b)


def _free_bytes(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_create_train_destroy_cycles_do_not_leak():
    """200 contexts created, fed one chunk, trained one batch epoch and one short online chunk, and destroyed: the device's
    free memory ends where it started (hipMemGetInfo; a few MiB of allocator slack allowed), and the last cycle still gives
    the first cycle's bits."""
    hip = C.CDLL("libamdhip64.so")                # the runtime libvsom_hip.so already runs on
    W, H, J, B = 24, 24, 48, 300
    X = gen.blobs(B, J, 5, 1, 2)
    init = gen.random_map(W * H, J, seed=42)

    def cycle():
        ctx = vsom_amd.Context(W, H, J, po.STANDARD)
        ctx.set_state(map=init)
        ctx.upload_chunk(X)
        mse = ctx.batch_epoch(5.0, True)
        ctx.upload_chunk(X[:40])
        ctx.train_online_chunk(0.05, 2.5, capi.EXPONENTIAL)
        st = ctx.get_state()
        ctx.close()
        return mse, st["map"]

    first = cycle()                                # (code objects, pools and the runtime's own caches come up here)
    cycle()
    before = _free_bytes(hip)
    for _ in range(200):
        last = cycle()
    after = _free_bytes(hip)
    assert before - after < 8 << 20, (before, after)
    assert beq(np.float32(first[0]), np.float32(last[0])) and beq(first[1], last[1])


def test_allocation_failure_is_nomem_and_the_context_survives():
    """vsom_upload_chunk / vsom_set_chunk_device of a chunk that cannot fit (2e9 rows x 784 values) return VSOM_ERR_NOMEM
    -- the allocation fails before a single row is read -- and the same context then trains a normal chunk to the
    oracle's bits."""
    W, H, J, B = 20, 20, 784, 400
    X = gen.mnist_like(B, seed=9, dim=J)
    init = (gen.random_map(W * H, J, seed=42) * np.float32(100)).astype(np.float32)
    ctx = vsom_amd.Context(W, H, J, po.STANDARD)
    ctx.set_state(map=init)
    ctx.upload_chunk(X)
    L = capi.lib()
    huge = 2_000_000_000
    fp = X.ctypes.data_as(C.POINTER(C.c_float))
    rc = L.vsom_upload_chunk(ctx._h, fp, C.c_size_t(huge))
    assert rc == -3, (rc, L.vsom_last_error())                         # VSOM_ERR_NOMEM
    assert b"hipMalloc" in L.vsom_last_error()
    dev_rows = L.vsom_device_ptr(ctx._h, capi.BUF_MAP)                 # any device pointer: never read
    rc = L.vsom_set_chunk_device(ctx._h, C.c_void_p(dev_rows), C.c_size_t(huge))
    assert rc == -3, (rc, L.vsom_last_error())
    with pytest.raises(capi.VsomError):                                # the failed staging left no chunk behind
        ctx.batch_epoch(4.0, True)
    o = po.OracleSom(W, H, J, po.STANDARD)
    o.set_state(map=init)
    for e, sigma in enumerate((6.0, 4.5)):
        ctx.upload_chunk(X)
        mse_g = ctx.batch_epoch(sigma, e == 0)
        lb = np.zeros(B, np.uint64)
        mse_o = o.batch_epoch(X, lb, sigma, e == 0, nthreads=8)
        assert beq(ctx.get_last_bmu(), lb) and beq(np.float32(mse_g), np.float32(mse_o))
    st = ctx.get_state()
    for k in ("map", "sigma", "weight", "hits"):
        assert beq(st[k], getattr(o, k)), k
    ctx.close()
