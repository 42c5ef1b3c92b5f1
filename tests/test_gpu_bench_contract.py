"""GPU: bench.py's contract -- ONE JSON line with the required keys, `roofline` and (N = 1) `cpu_baseline`;
`python bench.py --gpus 2` with no launcher starts its two ranks itself (here both on cuda:0 over gloo:
RCCL refuses one device twice) and times the real step, chunk replication included."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_small_shape():
    d = _run("--map", "32", "--chunk", "512", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1")
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "valu_fp32" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # both counts of the roofline: algorithmic (6 * N * D * B, the contract) and executed (live columns, 5 flop for
    # all-zero quads), plus the issue figures of the committed SQ counter pass (null when no profile of that key exists)
    assert 0 < r["frac_executed"] <= r["frac"] and r["executed_flop_per_launch"] <= r["algorithmic_flop_per_launch"]
    b = r["executed_basis"]
    assert 0 < b["live_columns"] <= 784 and b["column_quads"] == (b["live_columns"] + 3) // 4 and 0 < b["zero_quad_fraction"] < 1
    assert {"valu_issue_busy", "sustained_clock_ghz", "issue_source"} <= set(r)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    # `value` is quoted on the library's default arithmetic; the two contracted modes are timed beside it
    assert d["update_arithmetic"].startswith("strict") and d["backend"].startswith("none")
    assert [o["arithmetic"] for o in d["other_arithmetics"]] == ["sigma", "contracted"]
    assert all(o["value"] > 0 for o in d["other_arithmetics"])
    occ = d["config"]["column_occupancy"]
    assert 0 < occ["live"] <= occ["columns"] == 784
    assert "workload" in d["config"] and "model" not in d["config"]
    # the same step on rows that are not uint8-valued: normalised pixels and signed dense rows
    assert d["data_kind"]["name"] == "uint8_sparse"
    dv = {v["data"]: v for v in d["data_variants"]}
    assert set(dv) == {"float_sparse", "float_dense"}
    for v in dv.values():
        assert v["value"] > 0 and v["bmu_ms"] > 0 and v["update_ms"] > 0 and 0 < v["roofline"]["frac_executed"] <= v["roofline"]["frac"]
    assert dv["float_dense"]["live_columns"] == 784 and dv["float_sparse"]["live_columns"] == occ["live"]


def test_self_launch_two_ranks_on_one_device():
    d = _run("--gpus", "2", "--backend", "gloo", "--share-device", "--map", "32", "--chunk", "256", "--steps", "2",
             "--warmup", "1", "--no-cpu", "--no-other-arith")
    # gloo is NOT RCCL: the line says so and counts no RCCL rank
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["rccl_ranks"] == 0 and d["collective_ranks"] == 2
    assert d["backend"].startswith("gloo")
    # N > 1 quotes BASELINE config 3's split: the chunk of the step shared out over the ranks (strong) ...
    assert d["config"]["chunk_total"] == 256 and d["config"]["chunk_per_gpu"] == 128 and d["scaling"] == "strong"
    # ... and the weak variant (every rank owns a whole chunk) beside it
    w = d["weak_scaling"]
    assert w["chunk_total"] == 512 and w["chunk_per_gpu"] == 256 and w["value"] > 0
    assert "cpu_baseline" not in d                      # the CPU leg runs at N = 1 only
    assert d["value"] > 0 and d["roofline"]["avg_launch_ms"] > 0


def test_group_front_end_line():
    """--group: one process, vsom_group_* (three members rehearsed on device 0 with the peer transport)"""
    d = _run("--group", "--gpus", "3", "--share-device", "--map", "32", "--chunk", "384", "--steps", "2", "--warmup", "1")
    assert KEYS <= set(d) and d["n_gpus"] == 3 and d["scaling"] == "strong" and d["rccl_ranks"] == 0
    assert d["backend"].startswith("peer") and d["front_end"].startswith("vsom_group")
    assert d["config"]["chunk_total"] == 384 and d["config"]["chunk_per_gpu"] == 128 and d["value"] > 0


def test_online_line_is_bandwidth_priced():
    d = _run("--config", "online", "--map", "32", "--chunk", "64", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1")
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s" and d["value"] > 0
