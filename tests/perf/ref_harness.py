#!/usr/bin/env python3
"""The reference's own performance scenarios (tests/performance/perf_tests.cpp:74-402; the reference records no numbers
for them) on this build, through the C++ mirror of its API (host_api_test ref_harness), with the CPU oracle timed beside
(ref_harness_cpu.c, one thread).  One JSON line per scenario:
    python tests/perf/ref_harness.py [--scale 100] > profiles/r6_ref_harness.jsonl
Lives under tests/ because the CPU leg links the oracle (test infrastructure).  Needs a GPU."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HOST = os.path.join(ROOT, "variational-self-organizing-maps_amd", "host")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=100, help="the two million-call loops run 1e6 / scale calls")
    args = ap.parse_args()
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "ican_fixture.json")))
    rows = np.array(fx["rows"], np.float32)
    assert rows.shape == (20, 9), rows.shape
    tmp = tempfile.mkdtemp(prefix="vsom_refh_")
    fixture = os.path.join(tmp, "ican_rows.f32")
    rows.tofile(fixture)
    cpu_exe = os.path.join(tmp, "ref_harness_cpu")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libvsom_oracle.so"])
    subprocess.check_call(["gcc", "-O2", "-o", cpu_exe, os.path.join(HERE, "ref_harness_cpu.c"),
                           "-L" + os.path.join(ROOT, "oracle"), "-lvsom_oracle", "-lm", "-fopenmp",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    exe = os.path.join(HOST, "host_api_test")
    if not os.path.exists(exe):
        subprocess.check_call(["bash", os.path.join(HOST, "build.sh")], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.pop("VSOM_DEVICES", None)
    gpu = subprocess.run([exe, "ref_harness", fixture, str(args.scale)], capture_output=True, text=True, timeout=1500, env=env)
    if gpu.returncode != 0:
        sys.stderr.write(gpu.stdout + gpu.stderr)
        raise SystemExit(gpu.returncode)
    env["OMP_NUM_THREADS"] = "1"
    cpu = subprocess.run([cpu_exe, fixture, str(args.scale)], capture_output=True, text=True, timeout=1500, env=env)
    if cpu.returncode != 0:
        sys.stderr.write(cpu.stdout + cpu.stderr)
        raise SystemExit(cpu.returncode)
    cpu_by = {}
    for ln in cpu.stdout.splitlines():
        if ln.startswith("{"):
            d = json.loads(ln)
            cpu_by.setdefault(d["scenario"], {}).update(d)
    for ln in gpu.stdout.splitlines():
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        c = cpu_by.get(d["scenario"])
        unit = "us_per_epoch" if "us_per_epoch" in d else "us_per_call"
        d["mirror_" + unit] = d.pop(unit)
        if c:
            d["cpu_oracle_" + unit] = c["cpu_oracle_" + unit]
            d["mirror_over_cpu_time"] = round(d["mirror_" + unit] / max(c["cpu_oracle_" + unit], 1e-9), 3)
            if "cpu_oracle_faithful_" + unit in c:
                d["cpu_oracle_faithful_" + unit] = c["cpu_oracle_faithful_" + unit]
        print(json.dumps(d), flush=True)


if __name__ == "__main__":
    main()
