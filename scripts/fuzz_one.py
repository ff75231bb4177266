#!/usr/bin/env python3
"""scripts/fuzz_one.py SEED [ENV=VAL ...] -- one fuzz collection (tests/fuzz.py) through oracle/_ref/agc and through agc_amd/bin/agc_amd
with the given environment; prints whether the archives of every step agree (a debugging aid for tests/test_fuzz_archives.py)."""
import hashlib
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fuzz  # noqa: E402

seed = int(sys.argv[1])
env = dict(x.split("=", 1) for x in sys.argv[2:])
with tempfile.TemporaryDirectory() as td:
    case = fuzz.make_case(seed, os.path.join(td, "in"))
    ref_env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    want, _ = fuzz.run_case(os.path.join(ROOT, "oracle", "_ref", "agc"), case, td, "ref", threads="1", env=ref_env)
    got, errs = fuzz.run_case(os.path.join(ROOT, "agc_amd", "bin", "agc_amd"), case, td, "amd", env=dict(os.environ, **env))
    print("seed", seed, " ".join(case["args"] + case["carry"]), "steps", case["steps"], "env", env)
    for i, (w, g) in enumerate(zip(want, got)):
        print("  step", i, "same" if w == g else "DIFFERENT", None if g is None else hashlib.sha256(g).hexdigest()[:12] if isinstance(g, bytes) else g)
    if any(w != g for w, g in zip(want, got)):
        print(errs[-1][-1500:] if errs else "")
