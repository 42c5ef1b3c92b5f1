"""GPU: every kernel-selection switch the library reads from the environment (README "Switches") is exercised here, so
that no kernel in libvsom_hip.so is reachable only by hand:
  VSOM_NO_TINY=1            tiny maps through the general kernels instead of the one-workgroup batch epoch (vsom_tiny.hip)
                            and the one-launch online chunk (online_tiny_chunk_kernel: the online goldens then take the
                            per-sample kernels)
  VSOM_NO_CHAIN=1           small maps through the lane = node quad kernels instead of update_chain3_kernel
  VSOM_NO_COMPACT=1         no column compaction at all (the quad kernels on the full-width transposed chunk)
  VSOM_COMPACT_MIN_ROWS=1   every chunk compacted, however short
  VSOM_NO_DEDUPE=1          the exact search over every node, duplicate rows included (no representatives pass)
Each switch is read once per process, hence a fresh child interpreter per setting (a child process, never a re-exec),
running parity tests that compare the HIP path with the oracle bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(env_extra, files, kexpr=None, timeout=1200):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "gpu"]
    cmd += [os.path.join(ROOT, "tests", f) for f in files]
    if kexpr:
        cmd += ["-k", kexpr]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


def test_tiny_maps_through_the_general_kernels():
    _child({"VSOM_NO_TINY": "1"}, ["test_gpu_goldens.py", "test_gpu_batch_parity.py"])


def test_small_maps_through_the_quad_kernels():
    _child({"VSOM_NO_CHAIN": "1", "VSOM_NO_TINY": "1", "VSOM_ASM_SWEEP_N": "8"},
           ["test_gpu_goldens.py", "test_gpu_random_shapes.py", "test_gpu_batch_parity.py"])


def test_without_column_compaction():
    _child({"VSOM_NO_COMPACT": "1", "VSOM_ASM_SWEEP_N": "12"},
           ["test_gpu_random_shapes.py", "test_gpu_compact.py", "test_gpu_shortlist.py"])


def test_every_chunk_compacted():
    _child({"VSOM_COMPACT_MIN_ROWS": "1", "VSOM_ASM_SWEEP_N": "12"},
           ["test_gpu_random_shapes.py", "test_gpu_batch_parity.py", "test_gpu_shortlist.py", "test_gpu_group.py"])


def test_exact_search_without_the_duplicate_row_pass():
    _child({"VSOM_NO_DEDUPE": "1"}, ["test_gpu_dedupe.py", "test_gpu_shortlist.py"], kexpr="duplicate or redo or ties")
