"""GPU: the chain kernels whose workgroup shares ONE (c,w) stream through LDS (gen_update_asm.py, "lds") are chosen
by the library only for maps that fill the chip more than once (16384 nodes x 784 dims: tests at BASELINE's full C3
size cover them).  Here the SAME parity tests that the other kernels pass -- the assembly-kernel sweep with its
ragged depths, chunk tails, node shards, dead columns and zero runs, the shortest chunks, the compaction corner
cases -- run again in a child interpreter with VSOM_UPD_LDS=1, which forces those kernels onto every shape (the
switch is read once per process, hence the child)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suite_with_the_lds_sharing_kernels_forced():
    env = dict(os.environ)
    env["VSOM_UPD_LDS"] = "1"
    env["VSOM_ASM_SWEEP_N"] = env.get("VSOM_ASM_SWEEP_N", "24")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_random_shapes.py"),
                        os.path.join(ROOT, "tests", "test_gpu_compact.py"),
                        os.path.join(ROOT, "tests", "test_gpu_group.py"),
                        "-k", "assembly or shortest or compact or zero or dead or poisoned or group or live_set or nan_and_inf"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
