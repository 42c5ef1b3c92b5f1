"""GPU: double-buffered ingest (vsom_prefetch_chunk / vsom_commit_chunk).
Committing a prefetched chunk must leave exactly the state vsom_upload_chunk of the same data
leaves (staged rows, lastBMU zeroed -- DataSet.cpp:118-160), also when the next chunk's copy is
started while the current chunk is still training, from pinned and from pageable host memory."""
import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("tr,J", [(po.STANDARD, 24), (po.CLR, 6)])
def test_pipelined_chunks_match_oracle(pinned, tr, J):
    W = H = 12
    D = po.length(tr, J)
    sizes = [300, 128, 77, 300]                 # ragged, last one re-uses the first slot's size
    chunks = [gen.correlated(b, J, seed=10 + i) if tr == po.CLR else gen.blobs(b, J, 4, 1, 20 + i)
              for i, b in enumerate(sizes)]
    init = gen.random_map(W * H, D, seed=42)
    orc = po.OracleSom(W, H, J, tr)
    orc.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init)

    bufs = []
    def host(i):
        if not pinned:
            return np.ascontiguousarray(chunks[i])
        pb = capi.PinnedBuffer(chunks[i].shape)
        pb.array[...] = chunks[i]
        bufs.append(pb)
        return pb.array

    sigma = 5.0
    ctx.prefetch_chunk(host(0))
    for i in range(len(chunks)):
        ctx.commit_chunk()
        assert ctx.chunk_size == sizes[i]
        ctx.batch_epoch_async(sigma, i == 0)
        if i + 1 < len(chunks):
            ctx.prefetch_chunk(host(i + 1))     # copy of chunk i+1 beside the epoch of chunk i
        mse_g = ctx.get_mse()
        lb = np.zeros(sizes[i], np.uint64)      # every load zeroes lastBMU (DataSet.cpp:136-137)
        mse_o = orc.batch_epoch(chunks[i], lb, sigma, i == 0)
        assert (ctx.get_last_bmu() == lb).all(), i
        assert np.float32(mse_g) == np.float32(mse_o) or (np.isnan(mse_g) and np.isnan(mse_o)), i
        st = ctx.get_state()
        for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("weight", orc.weight)):
            same = (_bits(st[k]) == _bits(ref)) | (np.isnan(st[k]) & np.isnan(ref))
            assert same.all(), (i, k)
        assert (st["hits"] == orc.hits).all()
        sigma *= 0.9
    ctx.close()
    for b in bufs:
        b.free()


@pytest.mark.parametrize("source", ["host", "device"])
@pytest.mark.parametrize("tr,W", [(po.STANDARD, 48), (po.MEDIAN, 48), (po.STANDARD, 64), (po.STANDARD, 80)])
def test_chunks_staged_beside_the_running_epoch_match_oracle(source, tr, W):
    """On a map large enough for the lane = node chain kernels a prefetch that follows an asynchronous epoch also
    STAGES the next chunk -- on the copy stream, beside the chains of the current one (include/vsom_hip.h "Staging
    ahead"; lastBMU and the compaction's column record are double-buffered).  Chunks of different sizes, with and
    without dead columns / the compaction (>= 1024 rows), uint8-valued and float-valued, one larger than every buffer
    (staged at commit instead), first-epoch and local searches: every epoch is the oracle's bit for bit, and the
    current chunk's lastBMU / MSE stay readable between prefetch and commit."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")                # the runtime libvsom_hip.so already runs on (plain device buffers)
    # (W = 64: the 2200-row chunk is large enough for the G-less ring kernel of the search; W = 80 with rows of 784 values:
    #  more than two rounds of chain workgroups -- staging ahead is not offered there (csrc/vsom_update.hip): the chunk handed
    #  over ahead is copied beside the epoch and staged at its commit)
    H, J = W, (784 if W == 80 else 196)
    sizes = [1100, 1280, 77, 1100, 1500, 1024, 2200, 1100]
    kinds = ["u8", "u8", "u8", "f", "dense", "u8", "f", "u8"]
    chunks = []
    for i, (b, k) in enumerate(zip(sizes, kinds)):
        x = gen.mnist_like(b, seed=30 + i, dim=J)
        if k == "f":
            x = (x / np.float32(255)).astype(np.float32)
        elif k == "dense":
            x = gen.blobs(b, J, 5, 1, 40 + i, sigma=0.5)
        chunks.append(x)
    init = (gen.random_map(W * H, J, seed=42) * np.float32(100) + np.float32(100)).astype(np.float32)
    orc = po.OracleSom(W, H, J, tr)
    orc.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init)
    keep = []

    def give(i):
        if source == "host":
            pb = capi.PinnedBuffer(chunks[i].shape)
            pb.array[...] = chunks[i]
            keep.append(pb)
            ctx.prefetch_chunk(pb.array)
        else:
            x = np.ascontiguousarray(chunks[i])
            ptr = C.c_void_p()
            assert hip.hipMalloc(C.byref(ptr), C.c_size_t(x.nbytes)) == 0
            assert hip.hipMemcpy(ptr, x.ctypes.data_as(C.c_void_p), C.c_size_t(x.nbytes), C.c_int(1)) == 0   # host to device
            keep.append(ptr)
            ctx.stage_next_device(ptr.value, x.shape[0])

    sigma = 12.0
    give(0)
    for i in range(len(chunks)):
        ctx.commit_chunk()
        assert ctx.chunk_size == sizes[i]
        first = i in (0, 1, 4)                    # full searches and local ones
        ctx.batch_epoch_async(sigma, first)
        if i + 1 < len(chunks):
            give(i + 1)                           # copy AND staging of chunk i+1 beside the epoch of chunk i
        mse_g = ctx.get_mse()
        lb = np.zeros(sizes[i], np.uint64)
        mse_o = orc.batch_epoch(chunks[i], lb, sigma, first, nthreads=16)
        assert ctx.chunk_size == sizes[i]
        assert (ctx.get_last_bmu() == lb).all(), i
        assert np.float32(mse_g) == np.float32(mse_o) or (np.isnan(mse_g) and np.isnan(mse_o)), i
        st = ctx.get_state()
        for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("weight", orc.weight)):
            same = (_bits(st[k]) == _bits(ref)) | (np.isnan(st[k]) & np.isnan(ref))
            assert same.all(), (i, k)
        assert (st["hits"] == orc.hits).all()
        sigma *= 0.93
    # a chunk staged ahead and then overwritten by a plain upload: the upload wins, the prefetched chunk is staged at
    # its commit
    ctx.batch_epoch_async(sigma, False)           # (no reload: the search walks on from this chunk's last BMUs)
    orc.batch_epoch(chunks[-1], lb, sigma, False, nthreads=16)
    give(0)
    ctx.upload_chunk(chunks[2])
    lb = np.zeros(sizes[2], np.uint64)
    mse_o = orc.batch_epoch(chunks[2], lb, sigma, True, nthreads=16)
    mse_g = ctx.batch_epoch(sigma, True)
    assert np.float32(mse_g) == np.float32(mse_o) and (ctx.get_last_bmu() == lb).all()
    ctx.commit_chunk()
    lb = np.zeros(sizes[0], np.uint64)
    mse_o = orc.batch_epoch(chunks[0], lb, sigma, True, nthreads=16)
    mse_g = ctx.batch_epoch(sigma, True)
    assert np.float32(mse_g) == np.float32(mse_o) and (ctx.get_last_bmu() == lb).all()
    st = ctx.get_state()
    assert ((_bits(st["map"]) == _bits(orc.map)) | (np.isnan(st["map"]) & np.isnan(orc.map))).all()
    ctx.close()
    for b in keep:
        if hasattr(b, "free"):
            b.free()
        else:
            hip.hipFree(b)


def test_search_between_staged_ahead_prefetch_and_commit_is_refused():
    """Once the next chunk is staged ahead its rows sit in the staged-row buffers while B / lastBMU still describe the
    current chunk: every entry point that reads the rows must refuse (VSOM_ERR_INVALID) until vsom_commit_chunk -- also
    when the staged-ahead chunk has been abandoned for another one in the meantime -- and the committed chunk then
    trains like an uploaded one."""
    W = H = 48
    J = 196
    xs = [gen.mnist_like(1100, seed=70 + i, dim=J) for i in range(3)]
    init = (gen.random_map(W * H, J, seed=42) * np.float32(100) + np.float32(100)).astype(np.float32)
    orc = po.OracleSom(W, H, J, po.STANDARD)
    orc.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, po.STANDARD)
    ctx.set_state(map=init)
    bufs = []
    for x in xs:
        pb = capi.PinnedBuffer(x.shape)
        pb.array[...] = x
        bufs.append(pb)
    ctx.upload_chunk(xs[0])
    ctx.batch_epoch_async(10.0, True)
    ctx.prefetch_chunk(bufs[1].array)            # staged beside the chains of chunk 0
    lb = np.zeros(1100, np.uint64)
    orc.batch_epoch(xs[0], lb, 10.0, True, nthreads=16)
    assert (ctx.get_last_bmu() == lb).all()       # results of the current chunk stay readable
    for call in (ctx.bmu_batch, lambda: ctx.batch_epoch_async(9.0, False), lambda: ctx.train_online_chunk(0.1, 3.0, capi.EXPONENTIAL)):
        with pytest.raises(capi.VsomError, match="commit"):
            call()
    ctx.get_mse()                                 # (an entry point in between: the next prefetch cannot stage ahead)
    ctx.prefetch_chunk(bufs[2].array)            # abandons chunk 1, whose rows still sit in the buffers
    with pytest.raises(capi.VsomError, match="commit"):
        ctx.bmu_batch()
    ctx.commit_chunk()
    idx, _ = ctx.bmu_batch()
    mse_g = ctx.batch_epoch(9.0, True)
    lb = np.zeros(1100, np.uint64)
    mse_o = orc.batch_epoch(xs[2], lb, 9.0, True, nthreads=16)
    assert (idx == lb).all() and (ctx.get_last_bmu() == lb).all()
    assert np.float32(mse_g) == np.float32(mse_o)
    st = ctx.get_state()
    assert (_bits(st["map"]) == _bits(orc.map)).all() and (_bits(st["sigma"]) == _bits(orc.sigma)).all()
    ctx.close()
    for b in bufs:
        b.free()


def test_upload_without_the_wait():
    """vsom_upload_chunk_async: copy and staging enqueued on the context's stream, no wait (the first chunk of an epoch in the
    C++ mirror's online driver) -- the state it leaves is vsom_upload_chunk's: a batch epoch and an online chunk on chunks
    handed over that way are the oracle's, also when the next chunk follows without any synchronising call in between."""
    W, H, J = 20, 16, 24
    chunks = [gen.blobs(b, J, 5, 1, 50 + i, sigma=0.4) for i, b in enumerate((300, 41, 300))]
    init = gen.random_map(W * H, J, seed=42)
    orc = po.OracleSom(W, H, J, po.STANDARD)
    orc.set_state(map=init)
    ctx = vsom_amd.Context(W, H, J, po.STANDARD)
    ctx.set_state(map=init)
    bufs = []
    for x in chunks:
        pb = capi.PinnedBuffer(x.shape)
        pb.array[...] = x
        bufs.append(pb)
    # chunk 0 handed over and replaced by chunk 1 at once (nothing read it), then an epoch on chunk 1
    ctx.upload_chunk_async(bufs[0].array)
    ctx.upload_chunk_async(bufs[1].array)
    assert ctx.chunk_size == chunks[1].shape[0]
    lb = np.zeros(chunks[1].shape[0], np.uint64)
    mse_o = orc.batch_epoch(chunks[1], lb, 5.0, True)
    mse_g = ctx.batch_epoch(5.0, True)
    assert np.float32(mse_g) == np.float32(mse_o) and (ctx.get_last_bmu() == lb).all()
    # an online chunk with its results fetched in the same call, on a chunk handed over without a wait
    ctx.upload_chunk_async(bufs[2].array)
    lb = np.zeros(chunks[2].shape[0], np.uint64)
    run_o = orc.train_online_chunk(chunks[2], lb, 0.05, 3.0, capi.EXPONENTIAL)
    run_g, lb_g = ctx.train_online_chunk_fetch(0.05, 3.0, capi.EXPONENTIAL)
    assert np.float32(run_g) == np.float32(run_o) and (lb_g == lb).all()
    st = ctx.get_state()
    for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("S", orc.S), ("weight", orc.weight)):
        assert (_bits(st[k]) == _bits(ref)).all(), k
    assert (st["hits"] == orc.hits).all()
    ctx.close()
    for b in bufs:
        b.free()


def test_commit_without_prefetch_is_an_error():
    ctx = vsom_amd.Context(4, 4, 8)
    with pytest.raises(capi.VsomError):
        ctx.commit_chunk()
    ctx.close()


def test_online_chunk_mse_through_get_mse():
    """vsom_train_online_chunk with mse_out = NULL only enqueues; vsom_get_mse returns its MSE."""
    import ctypes as C
    W = H = 8
    J = 10
    X = gen.blobs(64, J, 3, 1, 2)
    init = gen.random_map(W * H, J, seed=42)
    a = vsom_amd.Context(W, H, J)
    a.set_state(map=init)
    a.upload_chunk(X)
    want = a.train_online_chunk(0.1, 3.0, capi.EXPONENTIAL)
    b = vsom_amd.Context(W, H, J)
    b.set_state(map=init)
    b.upload_chunk(X)
    capi.check(capi.lib().vsom_train_online_chunk(b._h, 0.1, 3.0, capi.EXPONENTIAL, None))
    got = b.get_mse()
    assert np.float32(got) == np.float32(want)
    a.close()
    b.close()
