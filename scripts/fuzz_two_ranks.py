"""The multi-GPU single-archive protocol against the live reference CLI on random collections (tests/fuzz.py): the files of a case go
through `oracle/_ref/agc create` and through tests/devsim/two_ranks_one_process (W compressors in one process on the CPU device
stand-in, prefetching schedule, two-step commit, writer-side bookkeeping queue; adaptive mode included) -- the archives must be
byte-identical.  Cases in -c mode are skipped (single-GPU only), and so are cases the device-sample API refuses as a whole (a contig
name twice inside one sample: the CLI path drops the second contig, AddSampleDevice is all or nothing).
usage: python scripts/fuzz_two_ranks.py [--from N] [--count M] [--ranks W] [--many] [--big]"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fuzz  # noqa: E402
from tests.devsim import build as simbuild  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--from", dest="first", type=int, default=0)
    ap.add_argument("--count", type=int, default=50)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--many", action="store_true")
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    ref = os.path.join(ROOT, "oracle", "_ref", "agc")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    tool = simbuild.build_two_ranks()
    bad = skipped = done = 0
    for seed in range(a.first, a.first + a.count):
        d = tempfile.mkdtemp(prefix=f"fz2_{seed}_")
        case = fuzz.make_case(seed, os.path.join(d, "in"), big=a.big, many=a.many)
        if "-c" in case["carry"]:
            skipped += 1
            shutil.rmtree(d)
            continue
        files = list(dict.fromkeys(case["files"]))  # (a file given twice is dropped by the CLI)
        opt = {"-k": "31", "-l": "20", "-s": "60000", "-b": "50"}
        for i in range(0, len(case["args"]) - 1):
            if case["args"][i] in opt:
                opt[case["args"][i]] = case["args"][i + 1]
        want = os.path.join(d, "ref.agc")
        r = subprocess.run([ref, "create"] + case["args"] + case["carry"] + ["-t", "1", "-o", want] + files, capture_output=True, env=env, timeout=600)
        got = os.path.join(d, "two.agc")
        t = subprocess.run([tool, got, opt["-k"], opt["-l"], opt["-s"], opt["-b"], "1" if "-a" in case["carry"] else "0"] + files,
                           capture_output=True, env=dict(os.environ, TWO_RANKS_W=str(a.ranks)), timeout=600)
        if t.returncode == 8 and b"already in the archive" in t.stderr:
            skipped += 1
            print(seed, "skipped (a sample with a repeated contig name)", flush=True)
        elif r.returncode != 0 or not os.path.exists(want):
            skipped += 1
            print(seed, "skipped (the reference CLI failed)", flush=True)
        else:
            same = t.returncode == 0 and os.path.exists(got) and open(got, "rb").read() == open(want, "rb").read()
            done += 1
            print(seed, "ok" if same else "MISMATCH", " ".join(case["args"] + case["carry"]), len(files), "files", "rc", t.returncode, flush=True)
            if not same:
                bad += 1
                print(t.stderr.decode(errors="replace")[-500:])
        shutil.rmtree(d)
    print(f"compared: {done}, skipped: {skipped}, mismatches: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
