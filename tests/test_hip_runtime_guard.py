"""capi.assert_single_hip_runtime: PyTorch's wheel bundles a libamdhip64 of its own (same SONAME as /opt/rocm's).  A
process that loads libvsom_hip.so first and imports torch afterwards holds two HIP runtimes; RCCL and torch streams handed
to the library then fail in obscure ways (tests/conftest.py).  The guard names the cause instead.  CPU-only: mapping the
libraries needs no device."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import sys
sys.path.insert(0, {root!r})
order = sys.argv[1]
if order == "torch_first":
    import torch
import vsom_amd
from vsom_amd import capi
capi.lib()
if order == "vsom_first":
    import torch
print("RUNTIMES", len(capi.hip_runtimes()))
try:
    capi.assert_single_hip_runtime("probe")
    print("GUARD ok")
except capi.VsomError as e:
    print("GUARD raised:", str(e)[:60])
"""


def _run(order):
    r = subprocess.run([sys.executable, "-c", PROBE.format(root=ROOT), order], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_torch_first_maps_one_runtime():
    out = _run("torch_first")
    assert "RUNTIMES 1" in out and "GUARD ok" in out, out


def test_vsom_first_then_torch_is_detected():
    out = _run("vsom_first")
    assert "RUNTIMES 2" in out and "GUARD raised: probe: two HIP runtimes" in out, out
