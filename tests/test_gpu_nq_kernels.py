"""GPU: the lane = (node, four dims) chain kernels (csrc/gen_nq_asm.py) are the library's choice for maps between the
small-map chain kernel and one resident round of lane = node wavefronts (BASELINE config 2 at full size, the node
shards of config 3's multi-GPU split: tests/test_gpu_baseline_configs.py, test_gpu_dist_ranks.py, test_gpu_group.py
reach them that way).  Here the SAME parity tests the other kernels pass -- the assembly-kernel sweep with its ragged
depths, chunk tails, node shards, dead columns and zero runs, the shortest chunks, the compaction corner cases, the
groups -- run again in a child interpreter with VSOM_UPD_NQ=1, which forces these kernels onto every shape the
lane = node kernels would take (the switch is read once per process, hence the child; a fresh interpreter, not a
re-exec); VSOM_UPD_NQ=0 is the opposite switch and gets a short run of the same sweep."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(env_extra, files, kexpr, timeout=1500):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"]
    cmd += [os.path.join(ROOT, "tests", f) for f in files]
    if kexpr:
        cmd += ["-k", kexpr]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


def test_parity_suite_with_the_node_quad_kernels_forced():
    _child({"VSOM_UPD_NQ": "1", "VSOM_ASM_SWEEP_N": os.environ.get("VSOM_ASM_SWEEP_N", "24")},
           ["test_gpu_random_shapes.py", "test_gpu_compact.py", "test_gpu_group.py", "test_gpu_batch_parity.py"],
           "assembly or shortest or compact or zero or dead or poisoned or group or live_set or nan_and_inf or batch")


def test_parity_sweep_with_the_node_quad_kernels_off():
    _child({"VSOM_UPD_NQ": "0", "VSOM_ASM_SWEEP_N": "12"}, ["test_gpu_random_shapes.py"], "assembly or shortest")
