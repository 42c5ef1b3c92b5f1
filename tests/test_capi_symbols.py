"""The C-ABI library loads and exports every symbol include/vsom_hip.h declares (no compute
calls: there is no GPU in the CPU test tier), and fails loudly -- not with a CPU fallback --
when no device is present."""
import ctypes
import os
import re

import pytest

import vsom_amd
from vsom_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "vsom_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vsom_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    L = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(capi.SYMBOLS) == names


def test_no_silent_cpu_fallback():
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(vsom_amd.VsomError):
        vsom_amd.Context(4, 4, 3)


def test_neighbourhood_weight_host_helper():
    from oracle import pyoracle as po
    for args in [(1, 0, 0, 0, 2.0), (3, 4, 0, 0, 2.0), (7, 2, 1, 9, 31.5), (2, 2, 2, 2, 1.0), (2, 3, 2, 2, 0.7)]:
        assert capi.neighbourhood_weight(*args) == po.neighbourhood_weight(*args)


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "variational-self-organizing-maps_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".sh")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in txt and "vsom_oracle" not in txt, os.path.join(dirpath, f)
