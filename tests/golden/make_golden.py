"""Generates tests/golden/*.npz from the REFERENCE itself (oracle/_ref, built in place from
/root/reference by oracle/Makefile).  Run in the authoring container only:

    python tests/golden/make_golden.py

Fixtures are data (inputs + the reference's outputs); no reference source is stored.
 * lz_golden.npz    -- (reference, text) -> CLZDiff_V2::Encode bytes, Estimate (several bounds),
                       GetCodingCostVector (both orientations)       [src/common/lz_diff.cpp]
 * scan_golden.npz  -- contig + splitter set -> accepted hit positions and k-mers, produced by the
                       reference's CKmer/bloom/hash-set driven as compress_contig does
                       [src/core/agc_compressor.cpp:2007-2036, src/core/kmer.h]
 * archives.json    -- sha256 + size of whole archives written by the reference CLI for the toy
                       collection (BASELINE.json configs[0]) and small synthetic collections
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from agc_amd import synth  # noqa: E402
from oracle import agc_oracle as O  # noqa: E402
from tests.cases import lz_cases  # noqa: E402


def lz_golden():
    cases = lz_cases(seed=2024, n_cases=48, lengths=(50, 200, 1000, 5000))
    d = {}
    for i, (mml, ref, text) in enumerate(cases):
        z = O.RefLZ(ref, mml)
        d[f"mml_{i}"] = np.array([mml], np.uint32)
        d[f"ref_{i}"] = ref
        d[f"text_{i}"] = text
        d[f"enc_{i}"] = z.encode(text)
        bounds = [0xFFFFFFFF, 0, 10, 100, text.size // 2]
        d[f"est_{i}"] = np.array([z.estimate(text, b) for b in bounds], np.uint64)
        d[f"cv0_{i}"] = z.cost_vector(text, 0)
        d[f"cv1_{i}"] = z.cost_vector(text, 1)
    d["n"] = np.array([len(cases)], np.uint32)
    np.savez_compressed(os.path.join(HERE, "lz_golden.npz"), **d)
    print("lz_golden.npz:", len(cases), "cases")


def scan_golden():
    rng = np.random.default_rng(77)
    d = {}
    n = 0
    for k in (17, 21, 25, 31, 32):
        refc = [synth.random_seq(rng, int(m)) for m in (30_000, 9_000)]
        spl = O.determine_splitters(refc, k, 1000)
        ctg = synth.mutate(rng, refc[0], 0.004, n_runs=3, iupac=4)
        pos, hd, hr = O.ref_scan_hits(ctg, k, spl)
        d[f"k_{n}"] = np.array([k], np.uint32)
        d[f"spl_{n}"] = spl
        d[f"ctg_{n}"] = ctg
        d[f"pos_{n}"], d[f"dir_{n}"], d[f"rc_{n}"] = pos, hd, hr
        n += 1
    d["n"] = np.array([n], np.uint32)
    np.savez_compressed(os.path.join(HERE, "scan_golden.npz"), **d)
    print("scan_golden.npz:", n, "cases")


def archives_append():
    """sha256 of the archives the reference CLI writes for create + append sequences (tests/collections.py APPEND_PLANS)"""
    from tests import collections as C
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib"))
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for plan in C.APPEND_PLANS:
            a = C.run_append_plan(O.REF_AGC, plan, os.path.join(td, plan + "_1"), threads="1", env=env)
            b = C.run_append_plan(O.REF_AGC, plan, os.path.join(td, plan + "_8"), threads="8", env=env)
            assert a == b, plan + ": the reference's append output depends on the thread count"
            out[plan] = [{"sha256": hashlib.sha256(x).hexdigest(), "size": len(x)} for x in a]
    json.dump(out, open(os.path.join(HERE, "archives_append.json"), "w"), indent=1)
    print("archives_append.json:", {k_: [x["size"] for x in v] for k_, v in out.items()})


def archives():
    """sha256 of the archives the reference CLI writes for tests/collections.py (t=1 and t=8 must agree)"""
    from tests import collections as C
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, (args, _kind) in C.CONFIGS.items():
            files = C.build(name, os.path.join(td, name))
            shas = []
            for t in ("1", "8"):
                fn = os.path.join(td, f"{name}_{t}.agc")
                subprocess.run([O.REF_AGC, "create"] + args + ["-t", t, "-o", fn] + files, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                b = open(fn, "rb").read()
                shas.append(hashlib.sha256(b).hexdigest())
            assert shas[0] == shas[1], f"{name}: reference output depends on the thread count"
            out[name] = {"args": args, "sha256": shas[0], "size": len(b)}
    out["zstd"] = "libzstd 1.4.9 (image's conda copy) on both sides"
    json.dump(out, open(os.path.join(HERE, "archives.json"), "w"), indent=1)
    print("archives.json:", {k_: v["size"] for k_, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    assert O.have_ref(), "oracle/_ref is not built (needs /root/reference)"
    lz_golden()
    scan_golden()
    archives()
    archives_append()
