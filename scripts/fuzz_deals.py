"""Random collections (tests/fuzz.py --many: 20-70 samples, small -b: packs fill again and again) through the N-rank protocol of
agc_amd/dist.py -- 2 or 3 processes over gloo on the CPU device stand-in (the worker of tests/test_dist_single_archive.py) -- with the
deals of full packs forced (AGC_AMD_DEAL_MIN_MB=0; a control step every sample or every third) against `oracle/_ref/agc create` on the
same files: byte-identical archives.  --no-deals: the same with deals off (Close deals everything), to tell the two apart.
usage: python scripts/fuzz_deals.py [--from N] [--count M] [--no-deals]"""
import argparse, os, shutil, socket, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _repeated_contig_name(files):
    for f in files:
        names = [l[1:].strip() for l in open(f, "rb").read().split(b"\n") if l.startswith(b">")]
        if len(set(names)) != len(names):
            return True
    return False


def main():
    import torch.multiprocessing as mp
    from tests import fuzz
    from tests import test_dist_single_archive as T
    from tests.devsim import build as simbuild
    ap = argparse.ArgumentParser()
    ap.add_argument("--from", dest="first", type=int, default=0)
    ap.add_argument("--count", type=int, default=20)
    ap.add_argument("--no-deals", action="store_true")
    a = ap.parse_args()
    simbuild.build()
    ref = os.path.join(ROOT, "oracle", "_ref", "agc")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    bad = done = 0
    for seed in range(a.first, a.first + a.count):
        d = tempfile.mkdtemp(prefix=f"deal{seed}_")
        case = fuzz.make_case(seed, os.path.join(d, "in"), many=True)
        args = case["args"] + case["carry"]
        files = list(dict.fromkeys(case["files"]))  # (a file given twice is dropped by the CLI)
        want_fn, got_fn = os.path.join(d, "ref.agc"), os.path.join(d, "dist.agc")
        r = subprocess.run([ref, "create"] + args + ["-t", "1", "-o", want_fn] + files, capture_output=True, env=env, timeout=600)
        if "-c" in args or r.returncode != 0 or not os.path.exists(want_fn):
            shutil.rmtree(d, ignore_errors=True)
            continue
        world = 2 + seed % 2
        os.environ["AGC_AMD_DEAL_MIN_MB"] = "-1" if a.no_deals else "0"
        os.environ["AGC_AMD_DEAL_EVERY"] = str(1 + 2 * (seed % 2))
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=T._worker, args=(r_, world, port, args, files, got_fn, q, False, True)) for r_ in range(world)]
        for p_ in ps:
            p_.start()
        for p_ in ps:
            p_.join(timeout=600)
        res = []
        while not q.empty():
            res.append(q.get())
        ok = os.path.exists(got_fn) and open(got_fn, "rb").read() == open(want_fn, "rb").read()
        if not ok and _repeated_contig_name(files):
            # (the CLI path drops the second contig of that name, the device-sample API is all or nothing: not a deal's business)
            print(seed, "skipped (a sample with a repeated contig name)", flush=True)
            shutil.rmtree(d, ignore_errors=True)
            continue
        done += 1
        print(seed, "ok" if ok else "MISMATCH", world, "ranks", " ".join(args), len(files), "files", os.path.getsize(want_fn),
              "deals", sorted({r_[7] for r_ in res if len(r_) > 7}), [r_[1] for r_ in res if r_[1] != "ok"], flush=True)
        bad += not ok
        shutil.rmtree(d, ignore_errors=True)
    print("cases:", done, "mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
