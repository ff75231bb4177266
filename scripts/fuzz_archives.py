"""Whole-archive parity fuzzer: random small collections (tests/fuzz.py) through the reference CLI (oracle/_ref/agc) and through
agc_amd -- the product CLI on a GPU box, or the host pipeline on the CPU device stand-in with --sim -- comparing every archive
byte for byte.  usage: python scripts/fuzz_archives.py [--sim] [--from N] [--count M] [--keep DIR]"""
import argparse
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fuzz  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sim", action="store_true")
    ap.add_argument("--from", dest="first", type=int, default=0)
    ap.add_argument("--count", type=int, default=50)
    ap.add_argument("--keep", default=None, help="directory that receives the inputs of failing cases")
    ap.add_argument("--many", action="store_true", help="20-70 samples per case (long speculation windows, many commit runs)")
    ap.add_argument("--big", action="store_true", help="Mbp-size contigs, segment sizes up to 1 M (32-bit LZ index regime)")
    a = ap.parse_args()
    ref = os.path.join(ROOT, "oracle", "_ref", "agc")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    if a.sim:
        from tests.devsim import build as simbuild
        cli = simbuild.build()
    else:
        from agc_amd import build
        build.build_host()
        cli = build.HOST_BIN
    bad = 0
    for seed in range(a.first, a.first + a.count):
        d = tempfile.mkdtemp(prefix=f"fuzz{seed}_")
        case = fuzz.make_case(seed, os.path.join(d, "in"), big=a.big, many=a.many)
        want, e1 = fuzz.run_case(ref, case, d, "ref", threads="1", env=env)
        # (every third seed: the whole-sample encode launched from the device's descriptors for every sample however small, every
        # registration a window of its own -- the path 3 Gbp samples take; every third: the LZ parses in 300-symbol chunks)
        amd_env = None
        if seed % 3 == 1:
            amd_env = dict(os.environ, AGC_AMD_DEV_ENCODE_MIN="0", AGC_AMD_WINDOW_MAX="1")
        elif seed % 3 == 2:
            amd_env = dict(os.environ, AGC_HIP_LZ_CHUNK="300")
        got, e2 = fuzz.run_case(cli, case, d, "amd", env=amd_env)
        # the reference itself dies on some inputs (e.g. `append -c` onto a partly filled batch): compare up to there
        n_cmp = len(want) - 1 if want and want[-1] is None else len(want)
        crashed = n_cmp != len(want)
        ok = want[:n_cmp] == got[:n_cmp] and all(x for x in want[:n_cmp]) and None not in got
        sz = lambda v: [None if x is None else len(x) for x in v]
        print(seed, ("ok" if ok else "MISMATCH") + (" (reference crashed at step %d)" % n_cmp if crashed else ""),
              " ".join(case["args"] + case["carry"]), case["steps"], sz(want), sz(got), flush=True)
        if not ok and None not in got:
            # a reference archive that does not even decode to its inputs is the reference's problem, not a parity gap
            i = next(j for j in range(n_cmp) if want[j] != got[j])
            used = case["files"][:sum(case["steps"][:i + 1])]
            if not fuzz.archive_round_trips(os.path.join(d, f"ref_{i}.agc"), used) and fuzz.archive_round_trips(os.path.join(d, f"amd_{i}.agc"), used):
                print("   step", i, ": the reference's own archive does not decode to its inputs (agc_amd's does) -- not counted")
                ok = True
        if not ok:
            bad += 1
            if e2[-1].strip():
                print("   stderr:", e2[-1][-400:])
            if a.keep:
                shutil.copytree(d, os.path.join(a.keep, f"case{seed}"), dirs_exist_ok=True)
        shutil.rmtree(d, ignore_errors=True)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
