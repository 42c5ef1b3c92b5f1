"""GPU: the single-process multi-GPU entry points of the C ABI (vsom_group_*, include/vsom_hip.h).

On the one-GPU test box:
  * a group of ONE device runs the whole orchestration over RCCL (ncclCommInitAll, the all-gathers as
    one-rank collectives) and must be bit-identical to vsom_batch_epoch and to the oracle;
  * groups of 2 / 3 / 4 members that all name device 0 exercise the N > 1 sharding, the ragged
    (broadcast-shaped) shards and the deferred sigmaMap / weightMap gathers -- RCCL refuses a device
    named twice, so these use the library's peer-copy transport -- bit-identical on EVERY member.
Som::trainBatchSomEpoch: Som.cpp:756-879 (samples shard at :764-782, nodes at :809-876)."""
import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def beq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


def data_for(tr, B, J):
    return gen.correlated(B, J, 5) if tr == capi.CLR else gen.blobs(B, J, 4, 1, 2)


def oracle_run(W, H, J, tr, X, init, sigmas):
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=init)
    mses, lbs = [], []
    for e, s in enumerate(sigmas):
        lb = np.zeros(X.shape[0], np.uint64)          # every load zeroes lastBMU (DataSet.cpp:136-137)
        mses.append(np.float32(o.batch_epoch(X, lb, s, e == 0)))
        lbs.append(lb)
    return o, mses, lbs


def group_run(g, X, init, sigmas, prefetch=False):
    g.set_state(map=init)
    mses, lbs = [], []
    for e, s in enumerate(sigmas):
        if prefetch:
            g.prefetch_chunk(X)
            g.commit_chunk()
            g.batch_epoch_async(s, e == 0)            # deferred gathers stay pending across epochs
            mses.append(g.get_mse())
        else:
            g.upload_chunk(X)
            mses.append(g.batch_epoch(s, e == 0))
        lbs.append(g.get_last_bmu().copy())
    g.synchronize()
    return mses, lbs


def check_members(g, o):
    for r in range(g.size):
        st = g.member(r).get_state()
        for k, ref in (("map", o.map), ("sigma", o.sigma), ("weight", o.weight), ("hits", o.hits)):
            assert beq(st[k], ref), (r, k)


@pytest.mark.parametrize("W,H,J,tr,B", [(16, 16, 48, capi.STANDARD, 300), (9, 7, 13, capi.MEDIAN, 70),
                                        (6, 6, 6, capi.CLR, 40), (32, 32, 784, capi.STANDARD, 512)])
def test_group_of_one_over_rccl_matches_single_context_and_oracle(W, H, J, tr, B):
    X = data_for(tr, B, J)
    D = capi.model_length(tr, J)
    init = gen.random_map(W * H, D, seed=42)
    sigmas = (5.0, 3.5, 2.0)
    g = vsom_amd.Group(W, H, J, tr, ndev=1)
    assert g.size == 1 and g.transport == "rccl"
    mses, lbs = group_run(g, X, init, sigmas)
    # the single-context path on the same inputs
    c = vsom_amd.Context(W, H, J, tr)
    c.set_state(map=init)
    for e, s in enumerate(sigmas):
        c.upload_chunk(X)
        assert beq(np.float32(c.batch_epoch(s, e == 0)), mses[e])
        assert beq(c.get_last_bmu(), lbs[e])
    a, b = g.get_state(), c.get_state()
    for k in ("map", "sigma", "weight", "hits"):
        assert beq(a[k], b[k]), k
    o, omse, olb = oracle_run(W, H, J, tr, X, init, sigmas)
    assert all(beq(m, n) for m, n in zip(mses, omse)) and all(beq(x, y) for x, y in zip(lbs, olb))
    check_members(g, o)
    c.close()
    g.close()


#        W   H   J  tr            B   members
CASES = [(16, 16, 48, capi.STANDARD, 300, 2),    # even shards
         (7, 5, 13, capi.MEDIAN, 70, 3),         # 35 nodes / 70 samples over 3: ragged shards
         (6, 6, 6, capi.CLR, 41, 2),             # CLR, odd sample count
         (32, 32, 784, capi.STANDARD, 512, 4),   # 784-dim rows, assembly update kernel on node shards
         (10, 10, 16, capi.STANDARD, 64, 4)]     # tiny map (the single context would take the one-launch epoch)


@pytest.mark.parametrize("W,H,J,tr,B,n", CASES)
@pytest.mark.parametrize("prefetch", [False, True])
def test_group_members_on_one_device_match_oracle(W, H, J, tr, B, n, prefetch):
    X = data_for(tr, B, J)
    D = capi.model_length(tr, J)
    init = gen.random_map(W * H, D, seed=42)
    sigmas = (5.0, 3.5, 2.0)
    g = vsom_amd.Group(W, H, J, tr, devices=[0] * n)
    assert g.size == n and g.transport == "peer"
    Xp = X
    pb = None
    if prefetch:
        pb = capi.PinnedBuffer(X.shape)
        pb.array[...] = X
        Xp = pb.array
    mses, lbs = group_run(g, Xp, init, sigmas, prefetch=prefetch)
    o, omse, olb = oracle_run(W, H, J, tr, X, init, sigmas)
    assert all(beq(m, n_) for m, n_ in zip(mses, omse))
    assert all(beq(x, y) for x, y in zip(lbs, olb))
    check_members(g, o)
    g.close()
    if pb is not None:
        pb.free()


def test_group_contracted_mode_and_empty_chunk():
    W, H, J, B = 16, 16, 48, 200
    X = gen.blobs(B, J, 4, 1, 2)
    init = gen.random_map(W * H, J, seed=42)
    g = vsom_amd.Group(W, H, J, capi.STANDARD, devices=[0, 0])
    g.set_update_mode(capi.UPDATE_FMA)
    g.set_state(map=init)
    g.upload_chunk(X)
    g.batch_epoch(4.0, True)
    c = vsom_amd.Context(W, H, J, capi.STANDARD)
    c.set_update_mode(capi.UPDATE_FMA)
    c.set_state(map=init)
    c.upload_chunk(X)
    c.batch_epoch(4.0, True)
    a, b = g.get_state(), c.get_state()
    for k in ("map", "sigma", "weight", "hits"):      # the same kernels on node shards: bit-identical
        assert beq(a[k], b[k]), k
    # an empty chunk still runs phase 2 and wipes the map (Som.cpp:840-875)
    g.upload_chunk(np.empty((0, J), np.float32))
    g.batch_epoch(4.0, False)
    st = g.get_state()
    assert (st["map"] == 0).all() and np.isnan(st["sigma"]).all() and (st["weight"] == 0).all()
    c.close()
    g.close()


def _random_group_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        tr = [capi.STANDARD, capi.MEDIAN, capi.CLR][i % 3]
        W, H = int(rs.randint(2, 40)), int(rs.randint(2, 40))
        J = int(rs.randint(2, 9)) if tr == capi.CLR else int(rs.choice([1, 3, 8, 17, 33, 64, 100, 200]))
        B = int(rs.choice([1, 2, 5, 16, 63, 64, 65, 130, 257]))
        members = int(rs.choice([2, 3, 4, 5]))
        out.append((f"g{i}_{['std', 'med', 'clr'][tr]}_{W}x{H}x{J}_B{B}_n{members}", W, H, J, tr, B, members,
                    float(rs.choice([1.5, 3.0, 7.0, 20.0])), int(rs.randint(1, 1 << 30))))
    return out


# VSOM_GROUP_SWEEP_N widens it for occasional long runs (default: 9 cases)
GROUP_CASES = _random_group_cases(int(__import__("os").environ.get("VSOM_GROUP_SWEEP_N", "9")), 20241004)


@pytest.mark.parametrize("name,W,H,J,tr,B,n,sigma,seed", GROUP_CASES, ids=[c[0] for c in GROUP_CASES])
def test_random_group_shapes_match_oracle(name, W, H, J, tr, B, n, sigma, seed):
    """random maps / chunks / member counts (more members than samples or nodes included: empty shards):
    first epoch + two local-search epochs, every member bit-identical to the oracle"""
    rs = np.random.RandomState(seed)
    X = (rs.randn(B, J) * rs.choice([0.1, 1.0, 50.0])).astype(np.float32)
    X[rs.rand(B, J) < 0.1] = 0.0
    init = gen.random_map(W * H, capi.model_length(tr, J), seed=seed % 1000)
    g = vsom_amd.Group(W, H, J, tr, devices=[0] * n)
    sigmas = (sigma, sigma * 0.8, sigma * 0.6)
    mses, lbs = group_run(g, X, init, sigmas)
    o, omse, olb = oracle_run(W, H, J, tr, X, init, sigmas)
    assert all(beq(a, b) for a, b in zip(mses, omse)), name
    assert all(beq(a, b) for a, b in zip(lbs, olb)), name
    check_members(g, o)
    g.close()


def test_borrowed_member_is_invalidated_when_the_group_closes():
    """Group.member() hands out a Context that wraps a vsom_ctx the group owns: it keeps the group alive, and
    after Group.close() a call on it raises instead of touching freed memory (r2 advisor finding)"""
    grp = capi.Group(6, 5, 7, capi.STANDARD, devices=[0, 0])
    m = grp.member(1)
    assert m.n_nodes == 30 and m.get_state(S=False)["map"].shape == (30, 7)
    grp.close()
    with pytest.raises(capi.VsomError):
        m.get_state()
    with pytest.raises(capi.VsomError):
        grp.member(0)
