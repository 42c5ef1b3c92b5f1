"""Multi-rank run of the REAL device engine (HipEngine over libvsom_hip.so) on one GPU box: the ranks
share cuda:0 and exchange through gloo (RCCL refuses two ranks on one device), so everything except
the transport is what bench.py --gpus N runs: sample-sharded phase 1, node-sharded phase 2, gathers on
the tensors that alias the library's device buffers, deferred sigma/weight gathers.  Every rank must
end bit-identical to the single-context epoch (Som.cpp:756-879) and to the oracle."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, case, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gen
        import vsom_amd
        vdist = importlib.import_module("variational-self-organizing-maps_amd.dist")
        W, H, J, tr, B, sigma = case
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        X = gen.correlated(B, J, 5) if tr == 2 else gen.blobs(B, J, 4, 1, 2)
        D = vsom_amd.capi.model_length(tr, J)
        init = gen.random_map(W * H, D, seed=42)
        stream = torch.cuda.Stream(device=dev)
        ctx = vsom_amd.Context(W, H, J, tr, device=0)
        ctx.set_state(map=init)
        ctx.set_stream(stream.cuda_stream)
        eng = vdist.HipEngine(ctx, dev, stream)
        trn = vdist.ShardedBatchTrainer(eng, rank, world)
        xt = torch.from_numpy(X).to(dev)
        out = {}
        for ep, first in enumerate((True, False, False)):
            with torch.cuda.stream(stream):
                eng.load_chunk_device(xt)
                trn.epoch(sigma, first)
                if ep == 1:
                    trn.flush()          # ep 0 -> 1 exercises the deferred gathers across epochs
            if ep != 0:
                torch.cuda.synchronize()
                out[f"lb{ep}"] = ctx.get_last_bmu().copy()
                out[f"mse{ep}"] = np.float32(ctx.get_mse())
        with torch.cuda.stream(stream):
            trn.flush()
        torch.cuda.synchronize()
        st = ctx.get_state()
        out.update(map=st["map"].copy(), sigma=st["sigma"].copy(), weight=st["weight"].copy(),
                   hits=st["hits"].copy())
        ret[rank] = out
        ctx.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


#        W   H   J  tr   B   sigma world
CASES = [(16, 16, 48, 0, 300, 5.0, 2),     # even node split, all_gather_into_tensor path
         (7, 5, 13, 1, 70, 2.5, 3),        # 35 nodes over 3 ranks: broadcast path, Median
         (6, 6, 6, 2, 40, 2.0, 2),         # CLR
         (32, 32, 784, 0, 512, 9.0, 4)]    # 784-dim rows, assembly update kernel on node shards


@pytest.mark.parametrize("case", CASES, ids=["even2", "uneven3-median", "clr2", "d784x4"])
def test_ranks_match_single_context(case):
    import torch.multiprocessing as mp
    import gen
    import vsom_amd
    *cfg, world = case
    W, H, J, tr, B, sigma = cfg
    ctxmp = mp.get_context("spawn")
    mgr = ctxmp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), tuple(cfg), ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == list(range(world))

    X = gen.correlated(B, J, 5) if tr == 2 else gen.blobs(B, J, 4, 1, 2)
    D = vsom_amd.capi.model_length(tr, J)
    ref = vsom_amd.Context(W, H, J, tr, device=0)
    ref.set_state(map=gen.random_map(W * H, D, seed=42))
    exp = {}
    for ep, first in enumerate((True, False, False)):
        ref.upload_chunk(X)
        exp[f"mse{ep}"] = np.float32(ref.batch_epoch(sigma, first))
        exp[f"lb{ep}"] = ref.get_last_bmu().copy()
    st = ref.get_state()
    ref.close()

    def beq(a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.dtype.kind == "f":
            return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()
        return (a == b).all()

    for r in range(world):
        got = ret[r]
        for ep in (1, 2):
            assert beq(got[f"lb{ep}"], exp[f"lb{ep}"]), (r, ep)
            assert beq(got[f"mse{ep}"], exp[f"mse{ep}"]), (r, ep, got[f"mse{ep}"], exp[f"mse{ep}"])
        for k in ("map", "sigma", "weight", "hits"):
            assert beq(got[k], st[k]), (r, k)


def test_rccl_world_of_one_on_the_aliased_buffers():
    """tools/dist_smoke.py: a world-size-1 NCCL (RCCL) process group whose collectives run on the zero-copy
    tensors that alias the library's device buffers -- the code path bench.py --gpus N takes on real peers"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_smoke.py")], capture_output=True, text=True,
                       timeout=300, env={**os.environ, "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert r.returncode == 0 and "dist smoke ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
