"""Random collections (tests/fuzz.py, --many: 20-70 samples, small -b: packs fill again and again) through `agc_amd.dist_create` with
two or three ranks sharing cuda:0 (gloo) and the deals of full packs forced (AGC_AMD_DEAL_MIN_MB=0, a control step every sample or
every third), against `oracle/_ref/agc create` on the same files: the archives must be byte-identical.
usage: python scripts/fuzz_deals_gpu.py [--from N] [--count M]"""
import argparse, hashlib, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import fuzz  # noqa: E402


def _repeated_contig_name(files):
    for f in files:
        names = [l[1:].strip() for l in open(f, "rb").read().split(b"\n") if l.startswith(b">")]
        if len(set(names)) != len(names):
            return True
    return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--from", dest="first", type=int, default=0)
    ap.add_argument("--count", type=int, default=20)
    a = ap.parse_args()
    ref = os.path.join(ROOT, "oracle", "_ref", "agc")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    bad = done = 0
    for seed in range(a.first, a.first + a.count):
        d = tempfile.mkdtemp(prefix=f"deal{seed}_")
        case = fuzz.make_case(seed, os.path.join(d, "in"), many=True)
        args = case["args"] + case["carry"]
        if "-c" in args:  # (the concatenated mode's units are dealt differently: covered by tests/test_dist_single_archive.py)
            shutil.rmtree(d, ignore_errors=True)
            continue
        files = list(dict.fromkeys(case["files"]))  # (a file given twice is dropped by the CLI)
        want_fn, got_fn = os.path.join(d, "ref.agc"), os.path.join(d, "dist.agc")
        r = subprocess.run([ref, "create"] + args + ["-t", "1", "-o", want_fn] + files, capture_output=True, env=env, timeout=600)
        if r.returncode != 0 or not os.path.exists(want_fn):
            shutil.rmtree(d, ignore_errors=True)
            continue
        world = 2 + seed % 2
        denv = dict(os.environ, AGC_AMD_DEAL_MIN_MB="0", AGC_AMD_DEAL_EVERY=str(1 + 2 * (seed % 2)))
        g = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                            "--master-port", str(29700 + seed % 200), "-m", "agc_amd.dist_create", "--backend", "gloo"] + args + ["-t", "4", "-o", got_fn] + files,
                           capture_output=True, text=True, env=denv, timeout=900, cwd=ROOT)
        ok = os.path.exists(got_fn) and open(got_fn, "rb").read() == open(want_fn, "rb").read()
        if not ok and _repeated_contig_name(files):
            print(seed, "skipped (a sample with a repeated contig name: the device-sample API is all or nothing)", flush=True)
            shutil.rmtree(d, ignore_errors=True)
            continue
        done += 1
        print(seed, "ok" if ok else "MISMATCH", world, "ranks", " ".join(args), len(files), "files", os.path.getsize(want_fn), flush=True)
        if not ok:
            bad += 1
            print("   stderr:", g.stderr[-600:])
        shutil.rmtree(d, ignore_errors=True)
    print("cases:", done, "mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
