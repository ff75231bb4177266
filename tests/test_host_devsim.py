"""Host logic of the create path on CPU: the unchanged host sources (agc_amd/csrc/host/) linked against the CPU device
stand-in of tests/devsim/ (include/agc_hip.h on top of the oracle) must write the reference's archives byte for byte.
This covers registration order, group/pack bookkeeping, adaptive mode, -c mode, zstd streams, collection metadata and
the container without a GPU; the HIP kernels themselves are covered by the `-m gpu` tests against the same goldens."""
import hashlib
import json
import os
import subprocess

import pytest

from tests import collections as C
from tests.devsim import build as simbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "archives.json")))


@pytest.fixture(scope="module")
def cli():
    return simbuild.build()


def _create(cli, args, files, out, threads="4"):
    r = subprocess.run([cli, "create"] + args + ["-t", threads, "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr[-2000:]
    return open(out, "rb").read()


@pytest.mark.parametrize("name", list(C.CONFIGS))
def test_host_pipeline_writes_the_reference_archive(cli, name, tmp_path):
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert len(got) == GOLD[name]["size"]
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


def test_host_pipeline_is_thread_independent(cli, tmp_path):
    args, _ = C.CONFIGS["syn_adaptive"]
    files = C.build("syn_adaptive", str(tmp_path / "in"))
    a = _create(cli, args, files, str(tmp_path / "a.agc"), threads="1")
    b = _create(cli, args, files, str(tmp_path / "b.agc"), threads="7")
    assert a == b


def test_product_library_is_not_the_simulator():
    """the product build must not pick the stand-in up: libagc_hip.so in agc_amd/ comes from hipcc and carries gfx950 code"""
    from agc_amd import build
    lib = build.build()
    assert os.path.dirname(lib) == os.path.join(ROOT, "agc_amd")
    blob = open(lib, "rb").read()
    assert b"gfx950" in blob and b"devsim" not in blob
