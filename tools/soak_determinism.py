#!/usr/bin/env python3
"""Race hunt: the same sequence of batch epochs (alternating chunks, first full-search then local
epochs, prefetch/commit ingest) run twice from the same start must end in bit-identical state; the
kernels contain no order-dependent reductions, so any difference is a synchronisation bug."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import vsom_amd  # noqa: E402
from vsom_amd import capi  # noqa: E402

W, D, B, steps = 128, 784, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 150
chunks = [gen.mnist_like(B, 3 + i, D) for i in range(3)]
init = gen.random_map(W * W, D, 42) * np.float32(100) + np.float32(100)


def run():
    ctx = vsom_amd.Context(W, W, D)
    ctx.set_state(map=init)
    pins = [capi.PinnedBuffer(c.shape) for c in chunks]
    for p, c in zip(pins, chunks):
        p.array[...] = c
    ctx.prefetch_chunk(pins[0].array)
    h = hashlib.sha256()
    for i in range(steps):
        ctx.commit_chunk()
        ctx.batch_epoch_async(40.0 * 0.99 ** i, i % 7 == 0)
        ctx.prefetch_chunk(pins[(i + 1) % 3].array)
        if i % 25 == 24:
            h.update(np.float32(ctx.get_mse()).tobytes())
            h.update(ctx.get_last_bmu().tobytes())
    st = ctx.get_state()
    for k in ("map", "sigma", "weight", "hits"):
        h.update(st[k].tobytes())
    ctx.close()
    return h.hexdigest()


a, b = run(), run()
print("run 1", a)
print("run 2", b)
assert a == b, "non-deterministic result"
print(f"deterministic over {steps} steps")
