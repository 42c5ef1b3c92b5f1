"""GPU: the opt-in contracted update arithmetic (VSOM_UPDATE_FMA) -- BMU indices, bmuHits, MSE and
weightMap stay bit-exact (they do not depend on the chain arithmetic); map / sigmaMap must be within
1e-5 relative fp32 of the oracle (BASELINE.json north_star tolerance), measured per model vector
against its largest magnitude."""
import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def rel_err(a, b):
    scale = np.maximum(np.abs(b).max(axis=1, keepdims=True), 1e-30)
    return float((np.abs(a.astype(np.float64) - b.astype(np.float64)) / scale).max())


@pytest.mark.parametrize("W,H,J,B,sigma,kind", [(24, 24, 784, 512, 8.0, "mnist"), (32, 32, 28, 1500, 10.0, "blobs"),
                                                (16, 16, 48, 4096, 6.0, "blobs"),
                                                # ragged depths: 16/14 column split (794), padded last slice (75)
                                                (40, 40, 794, 300, 9.0, "mnist"), (48, 48, 75, 700, 8.0, "blobs"),
                                                (128, 128, 784, 4096, 32.0, "mnist")])
def test_fma_mode_within_tolerance(W, H, J, B, sigma, kind):
    X = gen.mnist_like(B, 3, J) if kind == "mnist" else gen.blobs(B, J, 5, 1, 2, sigma=0.4)
    init = gen.random_map(W * H, J, 42) * (np.float32(100) if kind == "mnist" else np.float32(1))
    o = po.OracleSom(W, H, J)
    o.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    mse_o = o.batch_epoch(X, lb, sigma, True, nthreads=max(16, min(128, po.max_threads())))
    ctx = vsom_amd.Context(W, H, J)
    ctx.set_state(map=init)
    ctx.set_update_mode(capi.UPDATE_FMA)
    ctx.upload_chunk(X)
    mse_g = ctx.batch_epoch(sigma, True)
    st = ctx.get_state()
    assert (ctx.get_last_bmu() == lb).all()
    assert np.float32(mse_g) == np.float32(mse_o)
    assert (st["weight"].view(np.uint32) == o.weight.view(np.uint32)).all() and (st["hits"] == o.hits).all()
    assert rel_err(st["map"], o.map) <= RTOL
    assert rel_err(st["sigma"], o.sigma) <= RTOL
    # and it really is a different arithmetic (otherwise this test checks nothing)
    assert (st["map"].view(np.uint32) != o.map.view(np.uint32)).any()
    # strict mode on the same context is bit-identical again
    ctx.set_update_mode(capi.UPDATE_STRICT)
    ctx.set_state(map=init, hits=np.zeros(W * H, np.uint64))
    ctx.upload_chunk(X)
    ctx.batch_epoch(sigma, True)
    st = ctx.get_state()
    assert (st["map"].view(np.uint32) == o.map.view(np.uint32)).all()
    ctx.close()
