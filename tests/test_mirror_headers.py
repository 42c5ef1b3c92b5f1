"""CPU tests of the drop-in boundary's C++ mirror headers (host/include): what a caller written
against the reference's include/SOM.hpp & co. needs in order to recompile.

The reference CLI (apps/main.cpp) does not compile against the REFERENCE's own headers at this
snapshot: it uses `ARG_VERBOSE`, which include/SOM.hpp:17-35 never defines, and a
`Som{const char*, bool}` constructor SOM.hpp lacks (apps/main.cpp:43,193,205).  Against this
build's mirror exactly those faults -- and nothing else -- must remain.  The check reads
/root/reference and is therefore skipped where the reference is absent (the GPU box)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "variational-self-organizing-maps_amd", "host", "include")
REF = "/root/reference"


def _errors(src, std="c++23", extra=()):
    r = subprocess.run(["g++", f"-std={std}", "-fsyntax-only", "-I", INC, *extra, src],
                       capture_output=True, text=True)
    return [l for l in r.stderr.splitlines() if re.search(r"\berror\b", l)], r.stderr


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "apps", "main.cpp")), reason="reference tree absent")
def test_reference_cli_compiles_up_to_its_own_faults():
    errs, full = _errors(os.path.join(REF, "apps", "main.cpp"))
    kinds = set()
    for e in errs:
        if "ARG_VERBOSE" in e:
            kinds.add("ARG_VERBOSE")
        elif "Som::Som(<brace-enclosed initializer list>)" in e:
            kinds.add("Som{file,verbose}")
        else:
            kinds.add("OTHER: " + e)
    assert kinds == {"ARG_VERBOSE", "Som{file,verbose}"}, full
    # the two constructor calls (apps/main.cpp:193,205) and at least one ARG_VERBOSE use
    assert sum("Som::Som(<brace-enclosed" in e for e in errs) == 2


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "include", "SOM.hpp")), reason="reference tree absent")
def test_every_reference_macro_is_mirrored():
    """each #define of the reference's include/SOM.hpp:15-37 exists in the mirror with the same value"""
    pat = re.compile(r"^\s*#define\s+(\w+)\s+(\S+)\s*$", re.M)
    ref = dict(pat.findall(open(os.path.join(REF, "include", "SOM.hpp")).read()))
    ref.pop("_GLIBCXX_USE_C99", None)
    mine = dict(pat.findall(open(os.path.join(INC, "vsom_api.hpp")).read()))
    assert ref, "no macros found in the reference header"
    for k, v in ref.items():
        assert mine.get(k) == v, (k, v, mine.get(k))


def test_mirror_declares_reference_members(tmp_path):
    """a caller's translation unit that touches the members SOM.hpp:139-171 declares (incl.
    getSizeFromFile, the ARG_* argv indices and std::strcmp through the SOM.hpp include chain)"""
    src = tmp_path / "caller.cpp"
    src.write_text(r'''
#include "SOM.hpp"
#include "SqliteDataLoader.hpp"
int use(int argc, char **argv, Som &som, DataSet &data)
{
    if (argc > ARG_SOM_WEIGHT_DECAY_FUNCTION && std::strcmp(argv[ARG_SETTING], "-t") == 0) {
        som.randomInitialize(1, (float)std::atof(argv[ARG_SOM_INIT_SIGMA]));
        som.train(data, (size_t)std::atoi(argv[ARG_SOM_EPOCHS]), std::atof(argv[ARG_SOM_ETA0]),
                  std::atof(argv[ARG_SOM_ETA_DEC]), std::atof(argv[ARG_SOM_SIGMA0]), std::atof(argv[ARG_SOM_SIGMA_DEC]),
                  Som::WeigthDecayFunction::BatchMap);
        som.save(argv[ARG_SOM_FILE]);
    }
    Eigen::VectorXf v = som.getSizeFromFile(argv[ARG_SOM_FILE]);
    (void)argv[ARG_DB_FILE]; (void)argv[ARG_SOM_HEIGHT]; (void)argv[ARG_SOM_WIDTH];
    return som.measureSimilarity(&data, std::atoi(argv[ARG_ALLOWED_STD_DEV]), (size_t)std::atoi(argv[ARG_MIN_BMU_HITS]))
           + (int)v.size();
}
''')
    errs, full = _errors(str(src), std="c++20")
    assert not errs, full
