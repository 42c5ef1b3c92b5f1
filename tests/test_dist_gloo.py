"""world_size-2 gloo test of the sharded batch epoch (variational-self-organizing-maps_amd/dist.py)
on CPU: the same ShardedBatchTrainer that bench.py drives over RCCL, with an oracle-backed engine
in place of the HIP one.  Every rank must end bit-identical to the single-process epoch."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleEngine:
    """Engine interface of dist.py backed by the CPU oracle (tests only)."""

    def __init__(self, W, H, J, tr, init_map):
        from oracle import pyoracle as po
        self.o = po.OracleSom(W, H, J, tr)
        self.o.set_state(map=init_map)
        self.N = W * H
        self.map_rows = torch.from_numpy(self.o.map)
        self.sigma_rows = torch.from_numpy(self.o.sigma)
        self.weight = torch.from_numpy(self.o.weight)
        self._mse = None

    def load_chunk(self, X):
        self.X = np.ascontiguousarray(X, np.float32)
        self.B = self.X.shape[0]
        self._lb = np.zeros(self.B, np.uint64)
        self._sq = np.zeros(self.B, np.float32)
        self.lastbmu = torch.from_numpy(self._lb.view(np.int64))
        self.sqres = torch.from_numpy(self._sq)

    def phase1(self, s0, s1, is_first):
        self.o.batch_phase1_range(self.X, s0, s1, self._lb, self._sq, is_first)

    def finish(self):
        self._mse = self.o.batch_phase1_finish(self._lb, self._sq)

    def phase2(self, sigma, n0, n1):
        self.o.batch_phase2_range(self.X, self._lb, sigma, n0, n1)


def _worker(rank, world, port, case, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gen
        vdist = importlib.import_module("variational-self-organizing-maps_amd.dist")
        W, H, J, tr, B, sigma = case
        from oracle import pyoracle as po
        D = po.length(tr, J)
        X = gen.correlated(B, J, 5) if tr == 2 else gen.blobs(B, J, 4, 1, 2)
        init = gen.random_map(W * H, D, seed=42)
        eng = OracleEngine(W, H, J, tr, init)
        tr_ = vdist.ShardedBatchTrainer(eng, rank, world)
        out = {}
        for ep, first in enumerate((True, False)):
            eng.load_chunk(X)
            tr_.epoch(sigma, first)
            tr_.flush()
            out[f"lb{ep}"] = eng._lb.copy()
            out[f"mse{ep}"] = np.float32(eng._mse)
        out.update(map=eng.o.map.copy(), sigma=eng.o.sigma.copy(), weight=eng.o.weight.copy(),
                   hits=eng.o.hits.copy())
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CASES = [(8, 8, 10, 0, 37, 3.0),      # even node split, ragged sample split
         (7, 5, 6, 0, 20, 2.5),       # 35 nodes: uneven split -> broadcast path
         (6, 6, 4, 2, 16, 2.0)]       # CLR


@pytest.mark.parametrize("case", CASES, ids=["even", "uneven", "clr"])
def test_two_rank_epoch_matches_single_process(case):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]

    import gen
    from oracle import pyoracle as po
    W, H, J, tr, B, sigma = case
    X = gen.correlated(B, J, 5) if tr == 2 else gen.blobs(B, J, 4, 1, 2)
    o = po.OracleSom(W, H, J, tr)
    o.set_state(map=gen.random_map(W * H, po.length(tr, J), seed=42))
    exp = {}
    for ep, first in enumerate((True, False)):
        lb = np.zeros(B, np.uint64)
        exp[f"mse{ep}"] = o.batch_epoch(X, lb, sigma, first)
        exp[f"lb{ep}"] = lb

    def beq(a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.dtype.kind == "f":
            return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()
        return (a == b).all()

    for r in range(world):
        got = ret[r]
        for ep in (0, 1):
            assert beq(got[f"lb{ep}"], exp[f"lb{ep}"]) and beq(got[f"mse{ep}"], exp[f"mse{ep}"]), (r, ep)
        assert beq(got["map"], o.map) and beq(got["sigma"], o.sigma), r
        assert beq(got["weight"], o.weight) and beq(got["hits"], o.hits), r


def test_shard_bounds_cover():
    vdist = importlib.import_module("variational-self-organizing-maps_amd.dist")
    for total in (0, 1, 7, 64, 16384, 4097):
        for world in (1, 2, 3, 8):
            segs = [vdist.shard_bounds(total, world, r) for r in range(world)]
            assert segs[0][0] == 0 and segs[-1][1] == total
            assert all(segs[i][1] == segs[i + 1][0] for i in range(world - 1))
