#!/usr/bin/env python3
"""A/B of two code objects of the hand-scheduled chain kernels inside ONE process (development build of the library):
two contexts, each loading its own VSOM_ASM_HSACO, timed alternately; per-variant medians of the update launch.
  VSOM_LIB=tools/exp/bin/libvsom_dev.so python tools/exp/ab_hsaco.py tools/exp/bin/nt_base.hsaco tools/exp/bin/nt_cwl2.hsaco"""
import json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gen, vsom_amd

torch.cuda.set_device(0)
W = int(os.environ.get("AB_MAP", "128")); J = 784; B = 4096; sigma = W / 4.0
steps, rounds = int(os.environ.get("AB_STEPS", "20")), int(os.environ.get("AB_ROUNDS", "5"))
chunks = [torch.from_numpy(gen.mnist_like(B, seed=3 + i, dim=J)).cuda() for i in range(2)]
init = gen.random_map(W * W, J, 42) * np.float32(100) + np.float32(100)
ctxs = []
for path in sys.argv[1:3]:
    os.environ["VSOM_ASM_HSACO"] = os.path.abspath(path)
    c = vsom_amd.Context(W, W, J, 0)
    c.set_state(map=init)
    c.set_chunk_device(chunks[0].data_ptr(), B)
    c.batch_epoch_async(sigma, True)          # loads the code object named by the environment now
    c.synchronize()
    ctxs.append(c)
res = [[], []]
for r in range(rounds):
    for k in ((0, 1) if r % 2 == 0 else (1, 0)):
        c = ctxs[k]
        c.get_timing(reset=True)
        c.enable_timing(True, groups=["update"])
        for i in range(steps):
            c.set_chunk_device(chunks[i % 2].data_ptr(), B)
            c.batch_epoch_async(sigma, True)
        c.synchronize()
        tm = c.get_timing(reset=True)
        res[k].append(tm["update"][0] / tm["update"][1])
print(json.dumps({"map": W, "A": sys.argv[1], "B": sys.argv[2], "A_update_ms": [round(x, 4) for x in res[0]],
                  "B_update_ms": [round(x, 4) for x in res[1]], "A_median": round(float(np.median(res[0])), 4),
                  "B_median": round(float(np.median(res[1])), 4)}))
