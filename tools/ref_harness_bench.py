#!/usr/bin/env python3
"""The reference's own perf-harness scenarios (tests/performance/perf_tests.cpp:74-140,181-199,
390-401; it records no numbers) on the device path, with the CPU oracle timed beside it:
  1. Som::train, BatchMap, 10x10 map, the 20-row 9-dim fixture, 300 epochs (sigma0 10, decay 0.01
     -> stops when sigma < 1), time per epoch
  2. Som::train, Exponential, same data (eta0 0.001, decay 0.01), time per epoch
  3. trainSingle x1000, 100x100 map, depth 100, sigma 50, eta 0.1
  4. findBmu x1000, 100x100x100
These maps are tiny: the device path is bound by launch / host round-trip latency, not by any
roofline; the numbers say what a drop-in user of those scenarios would see."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import vsom_amd  # noqa: E402
from vsom_amd import capi, som  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

fx = json.load(open(os.path.join(ROOT, "tests", "golden", "ican_fixture.json")))
rows = np.array(fx["rows"], np.float32)


def report(name, gpu_s, cpu_s, unit):
    print(json.dumps({"scenario": name, "gpu_" + unit: round(gpu_s * 1e6, 2), "cpu_oracle_" + unit: round(cpu_s * 1e6, 2),
                      "gpu_over_cpu_time": round(gpu_s / cpu_s, 2)}), flush=True)


# 1/2: fixture training (the host loop is the Python mirror of Som::train)
for fn, label in ((som.WeigthDecayFunction.BatchMap, "train BatchMap 10x10x9, 20 rows"),
                  (som.WeigthDecayFunction.Exponential, "train Exponential 10x10x9, 20 rows")):
    init = gen.random_map(100, 9, 42)
    s = som.Som(10, 10, 9)
    s.setState(map=init)
    ds = som.ArrayDataSet(rows)
    s.train(ds, 3, 0.001, 0.01, 10.0, 0.01, fn)            # warm-up
    s.setState(map=init)
    t0 = time.perf_counter()
    s.train(ds, 300, 0.001, 0.01, 10.0, 0.01, fn)
    s.state()
    g = time.perf_counter() - t0
    o = po.OracleSom(10, 10, 9)
    o.set_state(map=init)
    t0 = time.perf_counter()
    if fn == som.WeigthDecayFunction.BatchMap:
        done, _ = o.train_batch(rows, [0, 20], 300, 10.0, 0.01)
    else:
        o.train_online(rows, [0, 20], 300, 0.001, 0.01, 10.0, 0.01, po.EXPONENTIAL)
        done = 300
    c = time.perf_counter() - t0
    report(label + f" ({done} epochs)", g / done, c / done, "us_per_epoch")
    if fn == som.WeigthDecayFunction.BatchMap:
        # the oracle's `faithful` mode keeps the reference's per-call temporaries and copies (what its
        # Eigen / std::function code does): closer to what the reference itself costs per epoch
        o2 = po.OracleSom(10, 10, 9)
        o2.set_state(map=init)
        lbf = np.zeros(20, np.uint64)
        t0 = time.perf_counter()
        for e in range(done):
            lbf[:] = 0
            o2.batch_epoch(rows, lbf, 10.0 * np.exp(-0.01 * e), e == 0, nthreads=1, faithful=True)
        cf = time.perf_counter() - t0
        report(label + " -- CPU side in the oracle's faithful (reference-allocation-pattern) mode", g / done, cf / done,
               "us_per_epoch")
    s.close()

# 3: trainSingle x1000 (perf_tests.cpp:114-140)
rs = np.random.RandomState(1)
init = gen.random_map(100 * 100, 100, 42)
v = rs.rand(1000, 100).astype(np.float32)
ctx = vsom_amd.Context(100, 100, 100)
ctx.set_state(map=init)
last = 0
for i in range(20):
    _, _, _, last = ctx.train_single(v[i], 0.1, 50.0, last, capi.EXPONENTIAL)
ctx.set_state(map=init)
last = 0
t0 = time.perf_counter()
for i in range(1000):
    _, _, _, last = ctx.train_single(v[i], 0.1, 50.0, last, capi.EXPONENTIAL)
g = time.perf_counter() - t0
o = po.OracleSom(100, 100, 100)
o.set_state(map=init)
last = 0
t0 = time.perf_counter()
for i in range(100):
    _, _, _, last = o.train_single(v[i], 0.1, 50.0, last, po.EXPONENTIAL)
c = (time.perf_counter() - t0) * 10
report("trainSingle x1000, 100x100x100, sigma 50 (one host call per sample)", g / 1000, c / 1000, "us_per_call")
# the same 1000 steps as one device-resident chunk (what trainBasicSom uses)
ctx.set_state(map=init)
ctx.upload_chunk(v)
ctx.train_online_chunk(0.1, 50.0, capi.EXPONENTIAL)
ctx.set_state(map=init)
ctx.upload_chunk(v)
t0 = time.perf_counter()
ctx.train_online_chunk(0.1, 50.0, capi.EXPONENTIAL)
g2 = time.perf_counter() - t0
report("same 1000 trainSingle steps enqueued as one chunk", g2 / 1000, c / 1000, "us_per_call")

# 4: findBmu x1000 (perf_tests.cpp:181-199)
ctx.set_state(map=init)
t0 = time.perf_counter()
for i in range(1000):
    ctx.find_bmu(v[i])
g = time.perf_counter() - t0
t0 = time.perf_counter()
for i in range(1000):
    o.find_bmu(v[i])
c = time.perf_counter() - t0
report("findBmu x1000, 100x100x100 (one host call per sample)", g / 1000, c / 1000, "us_per_call")
ctx.upload_chunk(v)
ctx.bmu_batch()
t0 = time.perf_counter()
ctx.bmu_batch()
g2 = time.perf_counter() - t0
report("same 1000 searches as one batched call", g2 / 1000, c / 1000, "us_per_call")
ctx.close()
