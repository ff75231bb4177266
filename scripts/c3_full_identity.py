"""BASELINE configs[2] at FULL contig size, asserted: one 3 Gbp GRCh38-shaped reference + N samples (d = 1e-3), -k 31 -l 15 -b 100.
agc_amd through the product path of bench.py (samples resident in the 2-bit layout, packed scan, GPU entropy stage) against the
reference CLI (oracle/_ref/agc) on the same data written as FASTA: the two archives must be byte-identical.

    python scripts/c3_full_identity.py [gbp=3.0] [n_samples=1] [sha256]   (needs a GPU, oracle/_ref/agc and ~7 GB per genome of scratch)

With a third argument -- the sha256 an earlier run of this script printed for the reference CLI's archive on the same sizes (the data are
seeded) -- the FASTA files and the 4-minute reference run are skipped: agc_amd's archive must have that hash.  AGC_IDENTITY_ANNOUNCE=1: every
next sample is announced (SetNextSamplePackedDevice), so its expansion + scan run ahead as in bench.py."""
import hashlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agc_amd import capi, host, synth, synth_dev

gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1
expect = sys.argv[3] if len(sys.argv) > 3 else None
announce = bool(os.environ.get("AGC_IDENTITY_ANNOUNCE"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "agc")
dev = torch.device("cuda:0")
total = int(gbp * 1e9)
ref, off = synth_dev.make_reference(total, 12345, dev)
tot = int(off[-1])
names = [f"chr{i + 1}" for i in range(len(off) - 1)]
scratch = "/dev/shm" if os.path.isdir("/dev/shm") else None
with tempfile.TemporaryDirectory(dir=scratch) as td:
    def fasta(path, t):
        if expect:
            return
        h = t[:tot].cpu().numpy()
        synth.to_fasta(path, [h[int(off[i]):int(off[i + 1])] for i in range(len(names))], names)
    t0 = time.time()
    files = [os.path.join(td, "ref.fa")]
    fasta(files[0], ref)
    samples = []
    for s in range(n_samples):
        smp = synth_dev.make_sample(ref, tot, 1e-3, 1000 + s, dev)
        samples.append(smp)
        files.append(os.path.join(td, f"s{s}.fa"))
        fasta(files[-1], smp)
    print(f"FASTA written in {time.time() - t0:.0f} s", flush=True)
    # ---- agc_amd: the API path bench.py times ----
    out_amd = os.path.join(td, "amd.agc")
    t0 = time.time()
    cmp_ = host.Compressor(0)
    cmp_.create(out_amd, 100, 31, None, 60000, 15, n_threads=16)
    cmp_.set_reference_dev(ref.data_ptr(), off)
    hctx = capi.Context.from_handle(cmp_.hip_ctx())
    pk, keep = hctx.pack_dev(ref, tot)
    cmp_.add_sample_packed_dev("ref", names, pk, off)
    packed = [hctx.pack_dev(smp, tot) for smp in samples]
    for s in range(len(samples)):
        if announce and s + 1 < len(samples):
            cmp_.set_next_sample_packed_dev(packed[s + 1][0], off)
        cmp_.add_sample_packed_dev(f"s{s}", names, packed[s][0], off)
    cmp_.close(16)
    st = cmp_.stats()
    cmp_.close_handle()
    print(f"agc_amd: {time.time() - t0:.1f} s, device entropy stage took {st['zstd_dev_in'] / 1e6:.0f} MB of {st['zstd_in'] / 1e6:.0f} MB", flush=True)
    a = open(out_amd, "rb").read()
    if expect:
        got = hashlib.sha256(a).hexdigest()
        print("agc_amd", len(a), got)
        print("expected (the reference CLI's archive, recorded)", expect)
        print("IDENTICAL" if got == expect else "DIFFERENT")
        sys.exit(0 if got == expect else 1)
    # ---- the reference CLI ----
    out_ref = os.path.join(td, "ref.agc")
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib"))
    t0 = time.time()
    subprocess.run([REF, "create", "-k", "31", "-l", "15", "-b", "100", "-t", "16", "-o", out_ref] + files, check=True, env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print(f"reference CLI: {time.time() - t0:.1f} s", flush=True)
    b = open(out_ref, "rb").read()
    print("agc_amd", len(a), hashlib.sha256(a).hexdigest())
    print("ref    ", len(b), hashlib.sha256(b).hexdigest())
    print("IDENTICAL" if a == b else "DIFFERENT")
    sys.exit(0 if a == b else 1)
