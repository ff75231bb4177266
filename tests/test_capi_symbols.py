"""CPU: the C-ABI library loads and exports every symbol include/agc_hip.h declares; without a
GPU every compute entry point fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "agc_hip.h")).read()
    return sorted(set(re.findall(r"\b(agc_hip_[a-z0-9_]+)\s*\(", h)))


def test_header_symbols_are_exported():
    from agc_amd import build, capi
    build.build()
    L = ctypes.CDLL(capi.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 25
    for s in decl:
        assert hasattr(L, s), f"{s} declared in include/agc_hip.h but not exported"
    assert sorted(capi.SYMBOLS) == decl, "agc_amd/capi.py binds a different symbol set than the header declares"
    assert L.agc_hip_abi_version() == 2


def test_no_device_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from agc_amd import capi
    with pytest.raises(capi.AgcHipError) as e:
        capi.Context(0)
    assert e.value.code == capi.ENODEV


def test_product_does_not_import_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    bad = []
    for dp, _dn, fn in os.walk(os.path.join(ROOT, "agc_amd")):
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|agc_oracle|libagc_oracle|oracle/_ref", t):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_compressor_header_matches_the_library_and_the_binding():
    """include/agc_cmp.h declares exactly the agc_cmp_* entry points libagc_host.so exports and agc_amd/host.py binds"""
    from agc_amd import build, host
    build.build_host()
    h = open(os.path.join(ROOT, "include", "agc_cmp.h")).read()
    decl = sorted(set(re.findall(r"\b(agc_cmp_[a-z0-9_]+)\s*\(", h)))
    src = open(os.path.join(ROOT, "agc_amd", "csrc", "host", "capi_host.cpp")).read()
    defined = sorted(set(re.findall(r"^(?:int|void|uint64_t|void \*|const char \*)\s*\*?(agc_cmp_[a-z0-9_]+)\(", src, re.M)))
    assert decl == defined
    bound = sorted(set(re.findall(r"L\.(agc_cmp_[a-z0-9_]+)\.", open(host.__file__).read())))
    assert bound == decl
    # loaded in a child process: the product's libagc_hip.so must not stay in this process, where other tests load the host
    # library linked against the CPU stand-in of the same soname
    import subprocess
    import sys
    code = ("import ctypes,sys\nL = ctypes.CDLL(sys.argv[1])\n"
            "missing = [s for s in sys.argv[2:] if not hasattr(L, s)]\nassert not missing, missing\n")
    subprocess.check_call([sys.executable, "-c", code, host.LIB_PATH] + decl)


def test_product_does_not_reach_into_tests():
    """the device stand-in (tests/devsim) and the fuzzer are test infrastructure: nothing under agc_amd/ may name them"""
    bad = []
    for dp, _dn, fn in os.walk(os.path.join(ROOT, "agc_amd")):
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"devsim|agc_hip_sim|tests\.", t):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
