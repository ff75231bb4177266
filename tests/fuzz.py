"""Random small collections for whole-archive parity fuzzing (tests/test_fuzz_archives.py, scripts/fuzz_archives.py).

Every case is derived from one integer seed: parameters (k, l, s, b, -c, -a), a reference with a handful of contigs and
samples built from it by the edits the reference's code paths care about -- substitutions, indels, N runs, IUPAC codes,
reverse-complemented contigs, rearranged / dropped / duplicated / novel contigs, contigs shorter than k, empty-ish
contigs, lower case -- and a split of the files into one `create` and zero or more `append` steps."""
import os

import numpy as np

from agc_amd import synth


def _rc(s):
    r = s[::-1].copy()
    m = r < 4
    r[m] = 3 - r[m]
    return r


def make_case(seed, outdir, big=False, many=False):
    """-> dict(args=[...], carry=[...], files=[...], steps=[n0, n1, ...]).
    big: Mbp-size contigs and segment sizes up to 1 M (references past 262 k symbols use the 32-bit LZ index, lz_diff.cpp:144-149);
    many: 20-70 samples per case"""
    rng = np.random.default_rng(seed)
    os.makedirs(outdir, exist_ok=True)
    k = int(rng.choice([17, 19, 21, 25, 31, 32]))
    l = int(rng.integers(15, min(k, 32) + 1)) if rng.random() < 0.7 else 20
    l = max(15, min(l, 32))
    s = int(rng.choice([100, 300, 1000, 2500, 60000]))
    if big:
        s = int(rng.choice([20000, 60000, 300000, 1000000]))
    b = int(rng.choice([1, 2, 3, 5, 50]))
    concat = rng.random() < 0.25
    adaptive = rng.random() < 0.3
    args = ["-k", str(k), "-l", str(l), "-s", str(s), "-b", str(b)]
    carry = (["-c"] if concat else []) + (["-a"] if adaptive else [])
    n_ctg = int(rng.integers(1, 6))
    scale = int(rng.choice([200, 2000, 20000]))
    if big:
        scale = int(rng.choice([100000, 400000]))
        n_ctg = int(rng.integers(1, 4))
    ref = [synth.random_seq(rng, int(rng.integers(1, 8)) * scale + int(rng.integers(0, 50))) for _ in range(n_ctg)]
    if rng.random() < 0.3:  # a contig shorter than k, and a repetitive one
        ref.append(synth.random_seq(rng, int(rng.integers(1, k))))
        unit = synth.random_seq(rng, int(rng.integers(2, 30)))
        ref.append(np.tile(unit, int(rng.integers(20, 400))))
    names = [f"c{i}" + (" extra text" if rng.random() < 0.3 else "") for i in range(len(ref))]
    files = []

    def write(fn, contigs, nm, lower=False):
        p = os.path.join(outdir, fn)
        synth.to_fasta(p, contigs, nm, width=int(rng.choice([60, 80, 100000])))
        if lower:
            with open(p, "rb") as f:
                lines = f.read().split(b"\n")
            with open(p, "wb") as f:
                f.write(b"\n".join(x if x.startswith(b">") else x.lower() for x in lines))
        files.append(p)

    write("ref.fa", ref, names)
    n_samples = int(rng.integers(1, 5 if big else 9))
    if many:  # long runs of small samples: speculation windows of 16-64 registrations, many commit runs and revalidations
        n_samples = int(rng.integers(20, 70))
    uniq = 0
    for si in range(n_samples):
        ctgs, nm = [], []
        order = rng.permutation(len(ref)) if rng.random() < 0.4 else np.arange(len(ref))
        for ci in order:
            if rng.random() < 0.12:
                continue
            c = ref[ci]
            d = float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3]))
            c = synth.mutate(rng, c, d, n_runs=int(rng.integers(0, 3)) if rng.random() < 0.4 else 0,
                             iupac=int(rng.integers(0, 4)) if rng.random() < 0.3 else 0,
                             indels=int(rng.integers(0, 4)) if rng.random() < 0.4 else 0)
            if rng.random() < 0.2:
                c = _rc(c)
            if rng.random() < 0.1 and c.size > 50:  # a fragment only
                a = int(rng.integers(0, c.size // 2))
                c = c[a:a + int(rng.integers(10, c.size - a))]
            if c.size == 0:
                continue
            ctgs.append(c)
            # in -c mode every contig is a sample named by its short name: keep them unique unless a clash is wanted
            if concat and rng.random() < 0.9:
                nm.append(f"s{si}_{uniq}")
                uniq += 1
            else:
                nm.append(names[ci] if rng.random() < 0.8 else f"s{si}_c{ci}")
        if rng.random() < 0.3:
            ctgs.append(synth.random_seq(rng, int(rng.integers(1, 6)) * scale))
            nm.append(f"novel{si}_{uniq}")
            uniq += 1
        # the same contig name twice inside one sample.  Not in -c mode: consecutive equal names are glued into one sample there
        # and the reference then drops the segments of later contigs (or dies), which is not behaviour worth restating
        dup = rng.random() < 0.1
        if dup and ctgs and not concat:
            ctgs.append(ctgs[0].copy())
            nm.append(nm[0])
        if concat:  # keep names unique inside the file
            seen = set()
            for i in range(len(nm)):
                while nm[i] in seen:
                    nm[i] += "x"
                seen.add(nm[i])
        if not ctgs:
            ctgs, nm = [synth.random_seq(rng, scale)], [f"only{si}"]
        write(f"x{si}.fa", ctgs, nm, lower=rng.random() < 0.15)
    if rng.random() < 0.1:  # the same file given twice (sanitize_input_file_names drops it)
        files.append(files[-1])
    n = len(files)
    steps = [n]
    if rng.random() < 0.5 and n > 1:
        cut = int(rng.integers(1, n))
        steps = [cut, n - cut]
        if rng.random() < 0.3 and n - cut > 1:
            c2 = int(rng.integers(1, n - cut))
            steps = [cut, c2, n - cut - c2]
    return {"args": args, "carry": carry, "files": files, "steps": steps}


def run_case(cli, case, outdir, tag, threads="3", env=None):
    """create + appends with `cli`; -> list of archive bytes (one per step; None where the program died), stderr texts"""
    import subprocess
    out, errs, pos, prev = [], [], 0, None
    for i, n in enumerate(case["steps"]):
        fn = os.path.join(outdir, f"{tag}_{i}.agc")
        if os.path.exists(fn):
            os.remove(fn)
        fl = case["files"][pos:pos + n]
        if i == 0:
            cmd = [cli, "create"] + case["args"] + case["carry"] + ["-t", threads, "-o", fn] + fl
        else:
            cmd = [cli, "append"] + case["carry"] + ["-t", threads, "-o", fn, prev] + fl
        r = subprocess.run(cmd, capture_output=True, timeout=300, env=env)
        errs.append(r.stderr.decode(errors="replace"))
        if r.returncode != 0:  # both CLIs exit with 0 even for user errors: anything else is a crash
            out.append(None)
            break
        out.append(open(fn, "rb").read() if os.path.exists(fn) else b"")
        pos += n
        prev = fn
    return out, errs


def archive_round_trips(archive_path, files):
    """True when every contig stored in the archive decodes to an input contig of that name (uses libagc_read.so).  Tells a
    reference archive that lost data (it happens in `append -c` onto a completely filled batch) from a genuine difference."""
    from agc_amd import build, reader
    build.build_read()
    inputs = {}
    for f in files:
        name, seq = None, []
        for line in open(f, "rb").read().split(b"\n"):
            if line.startswith(b">"):
                if name is not None:
                    inputs.setdefault(name, set()).add(b"".join(seq).upper())
                name, seq = line[1:].rstrip(b"\r").decode(), []
            else:
                seq.append(line.strip())
        if name is not None:
            inputs.setdefault(name, set()).add(b"".join(seq).upper())
    a = reader.CAGCFile()
    if not a.Open(archive_path):
        return False
    try:
        for sn in a.ListSample():
            ctgs = a.ListCtg(sn)
            if not ctgs:  # a sample without contigs cannot have been written on purpose
                return False
            for cn in ctgs:
                n = a.GetCtgLen(sn, cn)
                if n <= 0 or a.GetCtgSeq(sn, cn, -1, -1).encode() not in inputs.get(cn, ()):
                    return False
        return True
    finally:
        a.Close()


def read_side_matches(ref_cli, amd_cli, archive_path, seed, env=None, n_queries=6):
    """random getset / getctg (with and without ranges, by short name) / listctg queries: agc_amd's read side against the reference's
    on the same archive; returns the first differing query or None"""
    import subprocess
    from agc_amd import build, reader
    build.build_read()
    rng = np.random.default_rng(seed + 77)
    a = reader.CAGCFile()
    if not a.Open(archive_path):
        return "open"
    samples = a.ListSample()
    queries = [["listset", archive_path], ["listref", archive_path]]
    for _ in range(n_queries):
        sn = samples[int(rng.integers(0, len(samples)))]
        ctgs = a.ListCtg(sn)
        if not ctgs:
            continue
        cn = ctgs[int(rng.integers(0, len(ctgs)))]
        short = cn.split()[0]
        n = a.GetCtgLen(sn, cn)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            queries.append(["getset", "-l", str(int(rng.choice([40, 80, 1000]))), archive_path, sn])
        elif kind == 1:
            queries.append(["getctg", archive_path, f"{short}@{sn}"])
        elif kind == 2 and n > 2:
            lo = int(rng.integers(0, n - 1))
            hi = int(rng.integers(lo, min(n + 5, lo + 5000)))
            queries.append(["getctg", archive_path, f"{short}@{sn}:{lo}-{hi}"])
        else:
            queries.append(["listctg", archive_path, sn])
    a.Close()
    for q in queries:
        want = subprocess.run([ref_cli] + q, capture_output=True, timeout=120, env=env)
        got = subprocess.run([amd_cli] + q, capture_output=True, timeout=120)
        if want.returncode == 0 and want.stdout != got.stdout:
            return " ".join(q[:1] + q[2:])
    return None
