"""GPU: the contracted update arithmetic (VSOM_UPDATE_FMA; the arithmetic bench.py's `value` is quoted
on) against the oracle, ELEMENT BY ELEMENT.  Tolerance: BASELINE.json's north_star, 1e-5 relative fp32.

What does not depend on the chain arithmetic -- BMU indices, bmuHits, MSE, weightMap -- must stay
bit-exact.  For map / sigmaMap:
  * on the BASELINE workloads (MNIST-like rows, C2 64x64x784 and C3 128x128x784 at full size B=4096, and
    smaller shapes of the same data) EVERY element satisfies |a-b| <= 1e-5*|b| (measured <= 3.4e-7), and
    where the reference is exactly zero so is the result;
  * on signed data whose running means cancel towards zero (Gaussian blobs around the origin) no
    re-association of fp32 operations can bound the error of a cancelled element by its own magnitude;
    there every element satisfies |a-b| <= 1e-5*max(|b|, s_d), s_d = max_j |x_j,d| = the magnitude of the
    operands that chain consumed (measured <= 1e-7, i.e. below one ulp of s_d), every element of sigmaMap
    -- which has no cancellation -- still satisfies the pure element-wise bound, and the elements of map
    that miss the pure element-wise bound are all cancelled ones (|b| < 1e-2 * s_d).
Som.cpp:861-867 are the statements whose fp32 operations are contracted."""
import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def run_pair(W, H, J, B, sigma, kind):
    tr = capi.STANDARD
    X = gen.mnist_like(B, 3, J) if kind == "mnist" else gen.blobs(B, J, 5, 1, 2, sigma=0.4)
    init = gen.random_map(W * H, J, 42) * (np.float32(100) if kind == "mnist" else np.float32(1))
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    mse_o = o.batch_epoch(X, lb, sigma, True, nthreads=max(16, min(128, po.max_threads())))
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_state(map=init)
    ctx.set_update_mode(capi.UPDATE_FMA)
    ctx.upload_chunk(X)
    mse_g = ctx.batch_epoch(sigma, True)
    st = ctx.get_state()
    # bit-exact: everything that does not pass through the contracted chains
    assert (ctx.get_last_bmu() == lb).all()
    assert np.float32(mse_g) == np.float32(mse_o)
    assert (st["weight"].view(np.uint32) == o.weight.view(np.uint32)).all() and (st["hits"] == o.hits).all()
    # and it really is a different arithmetic (otherwise this test checks nothing)
    assert (st["map"].view(np.uint32) != o.map.view(np.uint32)).any()
    return X, init, o, ctx, st


def elementwise(a, b):
    """(max |a-b|/|b| over b != 0, #elements over RTOL, zero pattern identical)"""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    nz = b64 != 0
    rel = np.abs(a64 - b64)[nz] / np.abs(b64[nz])
    return (float(rel.max()) if rel.size else 0.0), int((rel > RTOL).sum()), bool((a64[~nz] == 0).all())


def strict_again(ctx, init, X, sigma, o, N):
    """strict mode on the same context is bit-identical again"""
    ctx.set_update_mode(capi.UPDATE_STRICT)
    ctx.set_state(map=init, hits=np.zeros(N, np.uint64))
    ctx.upload_chunk(X)
    ctx.batch_epoch(sigma, True)
    st = ctx.get_state()
    assert (st["map"].view(np.uint32) == o.map.view(np.uint32)).all()
    assert (st["sigma"].view(np.uint32) == o.sigma.view(np.uint32)).all()


@pytest.mark.parametrize("W,H,J,B,sigma", [(24, 24, 784, 512, 8.0),
                                           (40, 40, 794, 300, 9.0),      # ragged depth: 16/14 column split
                                           (64, 64, 784, 4096, 16.0),    # C2 at BASELINE's full size
                                           (128, 128, 784, 4096, 32.0)], # C3 at BASELINE's full size
                         ids=["24x24x784", "40x40x794", "C2_full", "C3_full"])
def test_contracted_mode_elementwise_on_baseline_workloads(W, H, J, B, sigma):
    X, init, o, ctx, st = run_pair(W, H, J, B, sigma, "mnist")
    for k, ref in (("map", o.map), ("sigma", o.sigma)):
        worst, over, zeros_ok = elementwise(st[k], ref)
        assert over == 0 and worst <= RTOL, (k, worst, over)
        assert zeros_ok, k                       # always-zero pixels: exactly zero in both
    strict_again(ctx, init, X, sigma, o, W * H)
    ctx.close()


@pytest.mark.parametrize("W,H,J,B,sigma", [(32, 32, 28, 1500, 10.0), (16, 16, 48, 4096, 6.0),
                                           (48, 48, 75, 700, 8.0),        # padded last slice
                                           (64, 64, 32, 16384, 16.0)],    # C4's shape, Standard: chain kernel
                         ids=["32x32x28", "16x16x48", "48x48x75", "64x64x32_B16384"])
def test_contracted_mode_on_signed_data_bounded_by_operand_scale(W, H, J, B, sigma):
    X, init, o, ctx, st = run_pair(W, H, J, B, sigma, "blobs")
    col = np.abs(X).max(axis=0, keepdims=True).astype(np.float64)             # s_d
    for k, ref in (("map", o.map), ("sigma", o.sigma)):
        a, b = st[k].astype(np.float64), ref.astype(np.float64)
        err = np.abs(a - b)
        assert (err <= RTOL * np.maximum(np.abs(b), col)).all(), (k, float((err / np.maximum(np.abs(b), col)).max()))
    # sigmaMap sums non-negative terms: no cancellation, so the pure element-wise bound holds for all of it
    worst, over, _ = elementwise(st["sigma"], o.sigma)
    assert over == 0, ("sigma", worst, over)
    # map: whatever misses the pure element-wise bound is a cancelled element
    a, b = st["map"].astype(np.float64), o.map.astype(np.float64)
    miss = np.abs(a - b) > RTOL * np.abs(b)
    assert (np.abs(b)[miss] < 1e-2 * np.broadcast_to(col, b.shape)[miss]).all()
    assert miss.mean() < 0.01
    strict_again(ctx, init, X, sigma, o, W * H)
    ctx.close()


def test_clr_has_one_arithmetic():
    """CombinatorialLinearRegression (Transformation.cpp:107-142): the recurrence feeds its own rounding
    back (inner = A*x' + B - y' depends on the accumulated A, B), and a fused variant measured 2e-5 of the
    node scale off the reference on a 12x12, J=9 map -- outside the tolerance -- so the library keeps CLR
    bit-identical whatever the mode says"""
    W, H, J, B, sigma = 12, 12, 9, 300, 3.0
    X = gen.correlated(B, J, 5)
    init = gen.random_map(W * H, capi.model_length(capi.CLR, J), 42)
    o = po.OracleSom(W, H, J, capi.CLR)
    o.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    o.batch_epoch(X, lb, sigma, True, nthreads=16)
    ctx = vsom_amd.Context(W, H, J, capi.CLR)
    ctx.set_state(map=init)
    ctx.set_update_mode(capi.UPDATE_FMA)
    ctx.upload_chunk(X)
    ctx.batch_epoch(sigma, True)
    st = ctx.get_state()
    assert (st["map"].view(np.uint32) == o.map.view(np.uint32)).all()
    same = (st["sigma"].view(np.uint32) == o.sigma.view(np.uint32)) | (np.isnan(st["sigma"]) & np.isnan(o.sigma))
    assert same.all()
    assert not capi.has_contracted(capi.CLR) and not capi.has_contracted(capi.MEDIAN) and capi.has_contracted(capi.STANDARD)
    ctx.close()


def test_median_is_bit_identical_in_both_modes():
    """the Median chains' fused operations are exact (gen_update_asm.py, compute_median): the mode changes
    nothing and the result stays bit-identical to the oracle"""
    W, H, J, B, sigma = 48, 48, 128, 600, 7.0
    X = gen.blobs(B, J, 5, 1, 2, sigma=0.4)
    init = gen.random_map(W * H, J, 42)
    o = po.OracleSom(W, H, J, capi.MEDIAN)
    o.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    o.batch_epoch(X, lb, sigma, True, nthreads=16)
    for mode in (capi.UPDATE_FMA, capi.UPDATE_STRICT):
        ctx = vsom_amd.Context(W, H, J, capi.MEDIAN)
        ctx.set_state(map=init)
        ctx.set_update_mode(mode)
        ctx.upload_chunk(X)
        ctx.batch_epoch(sigma, True)
        st = ctx.get_state()
        assert (st["map"].view(np.uint32) == o.map.view(np.uint32)).all()
        assert (st["sigma"].view(np.uint32) == o.sigma.view(np.uint32)).all()
        ctx.close()
