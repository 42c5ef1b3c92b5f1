"""Multi-GPU batch epoch: one process per GPU, torch.distributed (RCCL over xGMI on GPUs).

Exact-parity partitioning of Som::trainBatchSomEpoch (Som.cpp:756-879), see DESIGN.md:
  phase 1 (per-sample BMU search, :762-806)  -> shard SAMPLES, then all-gather lastBMU and the
          per-sample ||residual||^2 so that bmuHits / the fp32 MSE sum are formed in sample order
          on every rank;
  phase 2 (per-node sequential chains, :809-876) -> shard NODES (the variance accumulator uses
          the prefix mean, so sample-sharded partial sums cannot reproduce it), then all-gather
          the new map / sigmaMap / weightMap rows.
The result on every rank is bit-identical to the single-GPU epoch.

The trainer is written against a small "engine" interface so that the same orchestration is
exercised on CPU by the world_size-2 gloo tests (tests/test_dist_gloo.py, oracle-backed engine)
and on GPUs by bench.py (HipEngine below).
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist

from . import capi


def shard_bounds(total, world, rank):
    """Contiguous shard [lo, hi) of `total` items for `rank` of `world`."""
    return (total * rank) // world, (total * (rank + 1)) // world


class _CudaArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape),
                                         "typestr": typestr, "version": 2, "strides": None}


def device_tensor(ptr, shape, dtype, device):
    typestr = {torch.float32: "<f4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_CudaArray(ptr, shape, typestr), device=device)


class HipEngine:
    """vsom_ctx-backed engine: tensors alias the library's device buffers."""

    def __init__(self, ctx, device, stream=None):
        """`stream`: the torch.cuda.Stream both the library's kernels and the collectives run on (a new
        one when None).  The context adopts it (vsom_set_stream) and ShardedBatchTrainer makes it
        torch's current stream around every epoch()/flush(), so that kernels and collectives are
        ordered on ONE stream whatever stream the caller happens to be on."""
        # a torch stream / torch tensors handed to the library: both must live in ONE HIP runtime
        capi.assert_single_hip_runtime("dist.HipEngine")
        self.ctx = ctx
        self.device = torch.device(device)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        ctx.set_stream(self.stream.cuda_stream)
        self.N = ctx.n_nodes
        self.pitch = ctx.pitch
        self._bind_state()

    def stream_scope(self):
        return torch.cuda.stream(self.stream)

    def _bind_state(self):
        c, n, p, dev = self.ctx, self.N, self.pitch, self.device
        self.map_rows = device_tensor(c.device_ptr(capi.BUF_MAP), (n, p), torch.float32, dev)
        self.sigma_rows = device_tensor(c.device_ptr(capi.BUF_SIGMA), (n, p), torch.float32, dev)
        self.weight = device_tensor(c.device_ptr(capi.BUF_WEIGHT), (n,), torch.float32, dev)

    def load_chunk_device(self, x_tensor):
        """x_tensor: [B, J] fp32 on this GPU; staged on the context's stream."""
        assert x_tensor.is_contiguous() and x_tensor.dtype == torch.float32
        self.ctx.set_chunk_device(x_tensor.data_ptr(), x_tensor.shape[0])
        self._bind_chunk()

    def _bind_chunk(self):
        c, dev = self.ctx, self.device
        self.B = c.chunk_size
        # lastBMU is uint64 on the device; indices < 2^31, so an int64 view is value-identical
        self.lastbmu = device_tensor(c.device_ptr(capi.BUF_LASTBMU), (self.B,), torch.int64, dev)
        self.sqres = device_tensor(c.device_ptr(capi.BUF_SQRES), (self.B,), torch.float32, dev)

    def phase1(self, s0, s1, is_first):
        self.ctx.batch_phase1_async(s0, s1, is_first)

    def finish(self):
        self.ctx.batch_finish_async()

    def phase2(self, sigma, n0, n1):
        self.ctx.batch_phase2_async(sigma, n0, n1)

    def mse(self):
        return self.ctx.get_mse()


def _gather_rows(t, world, rank, group=None, async_op=False):
    """All ranks end up with every rank's contiguous row shard of `t` (dim 0), in place.
    With async_op the collective is only enqueued; returns what must be kept alive / waited on."""
    total = t.shape[0]
    if world == 1:
        return []
    pending = []
    if total % world == 0:
        lo, hi = shard_bounds(total, world, rank)
        # RCCL gathers in place (the input is this rank's slot of the output); gloo gets a copy
        src = t[lo:hi] if dist.get_backend(group) == "nccl" else t[lo:hi].clone()
        w = dist.all_gather_into_tensor(t, src, group=group, async_op=async_op)
        if async_op:
            pending.append((w, src))
    else:
        for r in range(world):
            lo, hi = shard_bounds(total, world, r)
            if hi > lo:
                w = dist.broadcast(t[lo:hi], src=r, group=group, async_op=async_op)
                if async_op:
                    pending.append((w, None))
    return pending


class ShardedBatchTrainer:
    """trainBatchSomEpoch over `world` ranks.  Each rank holds the whole chunk (the node-sharded
    phase 2 reads every sample) and the whole map (the sample-sharded phase 1 reads every node)."""

    def __init__(self, engine, rank=0, world=1, group=None):
        self.e, self.rank, self.world, self.group = engine, rank, world, group
        self._pending = []

    def _scope(self):
        # engines that own a device stream (HipEngine) run kernels AND collectives on it
        scope = getattr(self.e, "stream_scope", None)
        return scope() if scope is not None else contextlib.nullcontext()

    def flush(self):
        """Wait for the sigmaMap / weightMap gathers of the last epoch (call before reading state)."""
        with self._scope():
            for work, _keep in self._pending:
                work.wait()
        self._pending = []

    def epoch(self, sigma, is_first):
        with self._scope():
            self._epoch(sigma, is_first)

    def _epoch(self, sigma, is_first):
        e, w, r = self.e, self.world, self.rank
        s0, s1 = shard_bounds(e.B, w, r)
        e.phase1(s0, s1, is_first)
        _gather_rows(e.lastbmu, w, r, self.group)     # B x 8 B
        _gather_rows(e.sqres, w, r, self.group)       # B x 4 B
        e.finish()
        n0, n1 = shard_bounds(e.N, w, r)
        self.flush()                                  # phase 2 rewrites the rows the last gathers read
        e.phase2(sigma, n0, n1)
        _gather_rows(e.map_rows, w, r, self.group)    # N x pitch x 4 B: the next search needs it now
        # sigmaMap / weightMap are not read by the next phase 1: gather them behind it
        self._pending += _gather_rows(e.sigma_rows, w, r, self.group, async_op=True)
        self._pending += _gather_rows(e.weight, w, r, self.group, async_op=True)
