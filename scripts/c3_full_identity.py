"""BASELINE configs[2] at FULL contig size, asserted: one 3 Gbp GRCh38-shaped reference + N samples (d = 1e-3), -k 31 -l 15 -b 100.
agc_amd through the product path of bench.py (samples resident in HBM as the bytes of their FASTA files -> agc_hip_pack_fasta_* -> the
2-bit layout, packed scan, GPU entropy stage; AGC_IDENTITY_PREPACKED=1: packed from the codes, as before round 6) against the
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

    def packed_from_fasta(codes, what):
        """the bench's own front stage (bench.py: start_pack / packed_sample): the sample as the bytes of its FASTA file in HBM ->
        agc_hip_pack_fasta_begin / _end.  Every word of the 2-bit layout, the escape index and the contig offsets are compared with
        the pack of the codes themselves (agc_hip_pack_dev), at full size."""
        if os.environ.get("AGC_IDENTITY_PREPACKED"):
            return hctx.pack_dev(codes, tot), off
        raw, n_raw, rb, re_ = synth_dev.make_fasta(codes, off, names, int(os.environ.get("AGC_IDENTITY_FASTA_WIDTH", "60")))
        pk_f, keep_f, off_f = hctx.pack_fasta_dev(raw, n_raw, rb, re_)
        del raw
        assert np.array_equal(off_f, np.asarray(off, np.uint64)), "pack_fasta: contig offsets differ from the generator's"
        pk_c, keep_c = hctx.pack_dev(codes, tot)
        n_words, n_blocks = (tot + 15) // 16, (tot + 1023) // 1024
        wf, wc = keep_f[0][:n_words], keep_c[0][:n_words]
        if tot % 16:  # (bits beyond the last symbol of the last word are nobody's)
            m = (1 << (2 * (tot % 16))) - 1
            assert (int(wf[-1]) & m) == (int(wc[-1]) & m), "pack_fasta: last word differs"
            wf, wc = wf[:-1], wc[:-1]
        same_w = bool(torch.equal(wf, wc))
        same_i = bool(torch.equal(keep_f[1][:n_blocks], keep_c[1][:n_blocks]))
        print(f"pack_fasta of {what}: {n_raw} FASTA bytes -> {n_words} words; words {'==' if same_w else '!='} pack_dev(codes), "
              f"escape index {'==' if same_i else '!='}", flush=True)
        assert same_w and same_i, "pack_fasta: the packed sample differs from the pack of the codes"
        del pk_c, keep_c
        return (pk_f, keep_f), off_f

    (pk, keep), off_r = packed_from_fasta(ref, "the reference")
    cmp_.add_sample_packed_dev("ref", names, pk, off_r)
    packed = [packed_from_fasta(smp, f"sample {s}") for s, smp in enumerate(samples)]
    for s in range(len(samples)):
        if announce and s + 1 < len(samples):
            cmp_.set_next_sample_packed_dev(packed[s + 1][0][0], packed[s + 1][1])
        cmp_.add_sample_packed_dev(f"s{s}", names, packed[s][0][0], packed[s][1])
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
