#!/usr/bin/env python3
"""Single-GPU smoke of the RCCL code path: world_size-1 NCCL group, collectives forced to run on the
zero-copy tensors that alias the library's device buffers (the N>1 logic itself is covered by the
gloo tests)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
import gen, vsom_amd
from vsom_amd import capi
vdist = importlib.import_module("variational-self-organizing-maps_amd.dist")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W = H = 32; D = 48; B = 512
X = gen.blobs(B, D, 4, 1, 2); init = gen.random_map(W * H, D, 3)
stream = torch.cuda.Stream(device=dev)
ctx = vsom_amd.Context(W, H, D); ctx.set_state(map=init); ctx.set_stream(stream.cuda_stream)
eng = vdist.HipEngine(ctx, dev, stream)
xt = torch.from_numpy(X).to(dev)
def forced(t, world, rank, group=None, async_op=False):
    src = t.clone()
    w1 = dist.all_gather_into_tensor(t, src, group=group, async_op=async_op)   # world 1: gathers onto itself
    w2 = dist.broadcast(t, src=0, group=group, async_op=async_op)
    return [(w1, src), (w2, None)] if async_op else []
vdist._gather_rows = forced
tr = vdist.ShardedBatchTrainer(eng, 0, 1)
with torch.cuda.stream(stream):
    eng.load_chunk_device(xt)
    tr.epoch(5.0, True)
    tr.flush()
torch.cuda.synchronize()
ref = vsom_amd.Context(W, H, D); ref.set_state(map=init); ref.upload_chunk(X); ref.batch_epoch(5.0, True)
a, b = ctx.get_state(), ref.get_state()
assert all((a[k].view(np.uint8) == b[k].view(np.uint8)).all() for k in ("map", "sigma", "weight", "hits"))
assert (ctx.get_last_bmu() == ref.get_last_bmu()).all()
print("dist smoke ok: NCCL collectives on aliased buffers, results identical")
dist.destroy_process_group()
