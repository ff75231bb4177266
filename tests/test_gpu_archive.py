"""GPU end-to-end parity: `agc_amd create` (HIP kernels + host ordering contract + libzstd) must write
byte-identical archives to the reference CLI on the same inputs -- against the sha256 recorded from
the reference in tests/golden/archives.json, and against oracle/_ref/agc live when it was prebuilt."""
import hashlib
import json
import os
import subprocess

import pytest

from tests import collections as C

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AGC_AMD = os.path.join(ROOT, "agc_amd", "bin", "agc_amd")
REF_AGC = os.path.join(ROOT, "oracle", "_ref", "agc")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "archives.json")))


@pytest.mark.parametrize("name", list(C.CONFIGS))
def test_archive_bit_identical(name, tmp_path):
    from agc_amd import agc_container, build
    build.build_host()
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "amd.agc")
    r = subprocess.run([AGC_AMD, "create"] + args + ["-t", "8", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr[-2000:]
    got = open(out, "rb").read()
    if os.path.exists(REF_AGC):
        ref = str(tmp_path / "ref.agc")
        subprocess.run([REF_AGC, "create"] + args + ["-t", "4", "-o", ref] + files, check=True, capture_output=True, timeout=300)
        want = open(ref, "rb").read()
        assert hashlib.sha256(want).hexdigest() == GOLD[name]["sha256"], "reference build no longer matches its recorded golden"
        if got != want:
            pytest.fail("archive differs from the reference's:\n" + "\n".join(agc_container.diff(want, got)) + "\nstderr: " + r.stderr[-1500:])
    assert len(got) == GOLD[name]["size"], r.stderr[-1500:]
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


def test_cli_without_reference_file_reports_and_exits_zero(tmp_path):
    r = subprocess.run([AGC_AMD, "create", "-o", str(tmp_path / "x.agc"), str(tmp_path / "missing.fa")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Cannot" in r.stderr
