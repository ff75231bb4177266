"""GPU parity: splitter scan + preprocessing kernels vs the oracle (bit-exact)."""
import numpy as np
import pytest

from agc_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_hits(oracle, contigs, k, spl):
    ctg, pos, d, r = [], [], [], []
    for ci, c in enumerate(contigs):
        s = oracle.scan_contig(c, k, spl)
        m = s["back_full"] == 1
        ctg += [ci] * int(m.sum())
        pos += list((s["start"][m] + s["len"][m] - 1).astype(np.uint64))
        d += list(s["back_dir"][m])
        r += list(s["back_rc"][m])
    return np.array(ctg, np.uint32), np.array(pos, np.uint64), np.array(d, np.uint64), np.array(r, np.uint64)


@pytest.mark.parametrize("k", [17, 21, 25, 31, 32])
def test_scan_matches_oracle(hip_ctx, oracle, k):
    rng = np.random.default_rng(100 + k)
    refc = [synth.random_seq(rng, int(n)) for n in (70_000, 12_345, 150_001, 40, 5)]
    spl = oracle.determine_splitters(refc, k, 1000)
    assert spl.size > 50
    # sample: mutated copies + N-runs + IUPAC + contigs shorter than k + exact multiples of the 1 KiB wave step
    contigs = [synth.mutate(rng, refc[0], 0.002, n_runs=3, iupac=5),
               synth.mutate(rng, refc[1], 0.01),
               synth.mutate(rng, refc[2], 0.001, indels=3),
               refc[3].copy(), refc[4].copy(),
               refc[0][:65536].copy(), refc[0][:1024].copy(), refc[2][:4096 + k - 1].copy()]
    off = np.zeros(len(contigs) + 1, np.uint64)
    off[1:] = np.cumsum([c.size for c in contigs])
    codes = np.concatenate(contigs)
    hip_ctx.splitters_set(spl)
    assert hip_ctx.splitters_count() == spl.size
    got = hip_ctx.scan_contigs(codes, off, k)
    want = _oracle_hits(oracle, contigs, k, spl)
    for g, w, name in zip(got, want, ("ctg", "pos", "dir", "rc")):
        assert np.array_equal(g, w), name
    assert want[0].size > 100


def test_scan_dense_splitters_reset_rule(hip_ctx, oracle):
    # every k-mer of the contig is a splitter: the reset rule must keep exactly every k-th position
    rng = np.random.default_rng(5)
    k = 21
    c = synth.random_seq(rng, 20_000)
    allk = np.zeros(c.size, np.uint64)
    import ctypes as C
    n = oracle.lib().agco_enumerate_kmers(c.ctypes.data_as(C.POINTER(C.c_uint8)), c.size, k, allk.ctypes.data_as(C.POINTER(C.c_uint64)))
    spl = np.unique(allk[:n])
    hip_ctx.splitters_set(spl)
    got = hip_ctx.scan_contigs(c, [0, c.size], k)
    want = _oracle_hits(oracle, [c], k, spl)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
    assert np.all(np.diff(got[1].astype(np.int64)) == k)


def test_scan_empty_and_tiny(hip_ctx):
    hip_ctx.splitters_set(np.array([1, 2, 3], np.uint64))
    got = hip_ctx.scan_contigs(np.zeros(0, np.uint8), [0], 31)
    assert got[0].size == 0
    got = hip_ctx.scan_contigs(np.zeros(10, np.uint8), [0, 3, 3, 10], 31)
    assert got[0].size == 0


def test_splitters_insert(hip_ctx, oracle):
    rng = np.random.default_rng(9)
    k = 25
    c = synth.random_seq(rng, 50_000)
    spl = oracle.determine_splitters([c], k, 2000)
    half = spl[: spl.size // 2]
    hip_ctx.splitters_set(half)
    a = hip_ctx.scan_contigs(c, [0, c.size], k)
    hip_ctx.splitters_insert(spl[spl.size // 2:])
    b = hip_ctx.scan_contigs(c, [0, c.size], k)
    wa = _oracle_hits(oracle, [c], k, np.sort(half))
    wb = _oracle_hits(oracle, [c], k, spl)
    assert np.array_equal(a[1], wa[1]) and np.array_equal(b[1], wb[1]) and b[1].size > a[1].size


def test_preprocess(hip_ctx, oracle):
    import torch
    rng = np.random.default_rng(4)
    body = rng.choice(np.frombuffer(b"ACGTacgtNnRYSWKMBDHVUxz@`\n\r 0123>;*-", np.uint8), size=200_003)
    d_raw = torch.from_numpy(body).cuda()
    d_out = torch.zeros(body.size + 64, dtype=torch.uint8, device="cuda")
    n = hip_ctx.preprocess_dev(d_raw.data_ptr(), body.size, d_out.data_ptr())
    want = oracle.preprocess(body)
    assert n == want.size
    assert np.array_equal(d_out[:n].cpu().numpy(), want)


def _fasta_case(rng, contigs, width, eol=b"\n", lower=0.0, junk=0.0):
    """contigs (arrays of letters) -> (raw file bytes, raw_begin, raw_end): header lines between the bodies, `width` letters per
    line, optional lower-case letters and bytes < 64 sprinkled into the lines"""
    raw = bytearray()
    rb, re_ = [], []
    for i, c in enumerate(contigs):
        raw += b">ctg%d some description\n" % i
        rb.append(len(raw))
        c = c.copy()
        if lower:
            m = rng.random(c.size) < lower
            c[m] |= 0x20
        for a in range(0, c.size, width):
            line = bytes(c[a:a + width])
            if junk and rng.random() < junk:
                q = int(rng.integers(0, len(line) + 1))
                line = line[:q] + bytes(rng.choice(np.frombuffer(b" \t0123456789*-.;", np.uint8), size=int(rng.integers(1, 5)))) + line[q:]
            raw += line + eol
        re_.append(len(raw))
    return np.frombuffer(bytes(raw), np.uint8), np.array(rb, np.uint64), np.array(re_, np.uint64)


def _check_pack_fasta(hip_ctx, oracle, raw, rb, re_, esc_cap=64):
    import torch
    d_raw = torch.from_numpy(np.concatenate([raw, np.zeros(64, np.uint8)])).cuda()
    torch.cuda.synchronize()
    pk, keep, off = hip_ctx.pack_fasta_dev(d_raw, raw.size, rb, re_, esc_cap=esc_cap)
    want = [oracle.preprocess(raw[int(b):int(e)]) for b, e in zip(rb, re_)]
    woff = np.zeros(len(want) + 1, np.uint64)
    woff[1:] = np.cumsum([w.size for w in want])
    assert np.array_equal(off, woff), (off[:8], woff[:8])
    n = int(woff[-1])
    assert pk.n_symbols == n
    if n:
        out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        hip_ctx.expand_dev(pk, out.data_ptr())
        got = out[:n].cpu().numpy()
        exp = np.concatenate(want)
        if not np.array_equal(got, exp):
            bad = np.nonzero(got != exp)[0]
            raise AssertionError(f"{bad.size} symbols differ, first at {bad[:5]}: got {got[bad[:5]]} want {exp[bad[:5]]}")
        # the escaped blocks are exactly those that hold a symbol outside ACGT -- the packed form agc_hip_pack_dev gives for the
        # same codes (a clean block beside an N must not take an escape slot: ADVICE r5)
        nb = (n + 1023) // 1024
        pad = np.zeros(nb * 1024, np.uint8)
        pad[:n] = exp
        want_esc = (pad.reshape(nb, 1024) > 3).any(axis=1)
        got_esc = keep[1][:nb].cpu().numpy() >= 0
        assert np.array_equal(got_esc, want_esc), (np.nonzero(got_esc != want_esc)[0][:8], int(got_esc.sum()), int(want_esc.sum()))
    return pk, keep, off


@pytest.mark.parametrize("width,eol", [(60, b"\n"), (80, b"\r\n"), (17, b"\n"), (1, b"\n"), (100000, b"\n")])
def test_pack_fasta_matches_oracle_preprocess(hip_ctx, oracle, width, eol):
    """agc_hip_pack_fasta_dev (raw FASTA bodies -> the 2-bit layout in one pass) against the oracle's preprocess_raw_contig of every
    contig: symbol offsets and, through agc_hip_expand_dev, every symbol -- header lines between the bodies, several line widths and
    line ends, lower case, N runs and IUPAC codes (escaped blocks), stray bytes < 64, empty and tiny contigs"""
    rng = np.random.default_rng(5000 + width)
    letters = np.frombuffer(b"ACGT", np.uint8)
    sizes = [70_001, 0, 5, 16_384, 1023, 1024, 1025, 200_000 if width > 1 else 20_000, 1, 33_333]
    contigs = []
    for i, n in enumerate(sizes):
        c = rng.choice(letters, size=n)
        if i in (0, 7) and n > 5000:
            c[2000:2000 + 1500] = ord("N")                       # an N run: whole escaped blocks
            c[n // 2] = ord("R")
            c[n - 3] = ord("y")
            c[n // 3:n // 3 + 3] = np.frombuffer(b"@`U", np.uint8)  # cnv_num's odd corners
        contigs.append(c)
    raw, rb, re_ = _fasta_case(rng, contigs, width, eol, lower=0.3, junk=0.01)
    _check_pack_fasta(hip_ctx, oracle, raw, rb, re_, esc_cap=2)  # (2: the first attempt overflows, the binding grows the buffer)


def test_pack_fasta_edge_cases(hip_ctx, oracle):
    """no contigs, an empty buffer, ranges that skip most of the buffer, a body that is nearly all line ends (the read-ahead of a
    tile runs over several rounds), ranges that begin / end on tile and chunk boundaries, a contig of non-ACGT symbols only"""
    rng = np.random.default_rng(6001)
    letters = np.frombuffer(b"ACGT", np.uint8)
    z = np.zeros(0, np.uint64)
    _check_pack_fasta(hip_ctx, oracle, np.zeros(0, np.uint8), z, z)
    _check_pack_fasta(hip_ctx, oracle, rng.choice(letters, size=5000), z, z)
    body = rng.choice(letters, size=100_000)
    _check_pack_fasta(hip_ctx, oracle, body, np.array([0], np.uint64), np.array([body.size], np.uint64))
    _check_pack_fasta(hip_ctx, oracle, body, np.array([16384, 32768, 49152 + 16, 70_000, 100_000], np.uint64),
                      np.array([16384 + 16, 49152, 49152 + 32, 70_000, 100_000], np.uint64))
    sparse = np.full(300_000, 10, np.uint8)
    sparse[rng.choice(sparse.size, size=4000, replace=False)] = ord("G")
    sparse[150_000:151_100] = rng.choice(letters, size=1100)
    _check_pack_fasta(hip_ctx, oracle, sparse, np.array([0, 200_000], np.uint64), np.array([200_000, 300_000], np.uint64))
    only_n = np.full(50_000, ord("N"), np.uint8)
    _check_pack_fasta(hip_ctx, oracle, only_n, np.array([3], np.uint64), np.array([49_999], np.uint64))


def test_pack_fasta_many_tiny_contigs(hip_ctx, oracle):
    """20 000 contigs of 1-400 letters (a fragmented assembly: several contigs per 16-byte chunk, hundreds per tile): every range
    boundary falls somewhere else in a chunk, headers are longer than the bodies"""
    import time
    rng = np.random.default_rng(6200)
    letters = np.frombuffer(b"ACGTacgtN", np.uint8)
    contigs = [rng.choice(letters, size=int(n)) for n in rng.integers(1, 400, size=20_000)]
    contigs[7] = np.zeros(0, np.uint8)
    raw, rb, re_ = _fasta_case(rng, contigs, 70)
    t0 = time.time()
    _check_pack_fasta(hip_ctx, oracle, raw, rb, re_, esc_cap=4096)
    assert time.time() - t0 < 60


def test_pack_fasta_equals_preprocess_and_pack_on_a_big_sample(hip_ctx, oracle):
    """size-independent property at a larger size (120 MB of FASTA, 9 contigs): the one-pass kernel gives what the three-pass
    preprocess + the packing of its codes give, and the packed scan reports the same hits on both"""
    import torch
    rng = np.random.default_rng(6100)
    letters = np.frombuffer(b"ACGT", np.uint8)
    ref = rng.choice(letters, size=2_000_000)
    contigs = []
    for i in range(9):
        c = np.tile(ref, 6)[: 11_000_000 + 77_777 * i].copy()
        m = rng.random(c.size) < 0.001
        c[m] = rng.choice(letters, size=int(m.sum()))
        if i % 3 == 0:
            c[5_000_000:5_003_000] = ord("N")
        contigs.append(c)
    raw, rb, re_ = _fasta_case(rng, contigs, 60)
    pk, keep, off = _check_pack_fasta(hip_ctx, oracle, raw, rb, re_)
    codes = np.concatenate([oracle.preprocess(raw[int(b):int(e)]) for b, e in zip(rb, re_)])
    d = torch.from_numpy(codes).cuda()
    torch.cuda.synchronize()
    pk2, keep2 = hip_ctx.pack_dev(d)
    k = 31
    spl = oracle.determine_splitters([oracle.preprocess(ref)], k, 20_000)
    hip_ctx.splitters_set(spl)
    a = hip_ctx.scan_packed_dev(pk, off, k, cap=1 << 18)
    b = hip_ctx.scan_packed_dev(pk2, off, k, cap=1 << 18)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[0].size > 1000


@pytest.mark.parametrize("k,seg", [(31, 1000), (21, 500), (17, 3000), (32, 700)])
def test_determine_splitters_matches_oracle(hip_ctx, oracle, k, seg):
    """reference preprocessing on the GPU (agc_hip_determine_splitters_dev) vs the oracle's restatement of
    determine_splitters: duplicated regions (non-singleton k-mers), N-runs, contigs shorter than k, a contig
    that is one long repeat (no singleton at all)"""
    import torch
    rng = np.random.default_rng(200 + k)
    a = synth.random_seq(rng, 40_000)
    b = synth.random_seq(rng, 25_000)
    b[5_000:9_000] = a[10_000:14_000]                       # shared region: those k-mers are not singletons
    c = synth.mutate(rng, synth.random_seq(rng, 12_000), 0, n_runs=4, iupac=3)
    d = np.tile(synth.random_seq(rng, 50), 100)             # pure repeat
    e = synth.random_seq(rng, k - 1)                        # shorter than k
    f = oracle.rev_comp(a[20_000:26_000])                   # reverse-complement copy: same canonical k-mers
    contigs = [a, b, c, d, e, f]
    off = np.zeros(len(contigs) + 1, np.uint64)
    off[1:] = np.cumsum([x.size for x in contigs])
    dev = torch.from_numpy(np.concatenate(contigs)).cuda()
    got, srt = hip_ctx.determine_splitters_dev(dev.data_ptr(), off, k, seg, want_sorted=True)
    want = oracle.determine_splitters(contigs, k, seg)
    assert np.array_equal(got, want), (got.size, want.size)
    assert want.size > 20
    # the sorted k-mer list (adaptive mode) = all canonical k-mers of the reference
    import ctypes as C
    allk = []
    for x in contigs:
        buf = np.zeros(max(x.size, 1), np.uint64)
        n = oracle.lib().agco_enumerate_kmers(x.ctypes.data_as(C.POINTER(C.c_uint8)), x.size, k, buf.ctypes.data_as(C.POINTER(C.c_uint64)))
        allk.append(buf[:n])
    assert np.array_equal(srt, np.sort(np.concatenate(allk)))


# ---- 2-bit packed samples ---------------------------------------------------------------------------------------------
def _packed_case(oracle, rng, k):
    refc = [synth.random_seq(rng, int(n)) for n in (70_000, 12_345, 150_001, 40, 5)]
    spl = oracle.determine_splitters(refc, k, 1000)
    contigs = [synth.mutate(rng, refc[0], 0.002, n_runs=3, iupac=5),
               synth.mutate(rng, refc[1], 0.01),
               synth.mutate(rng, refc[2], 0.001, indels=3),
               refc[3].copy(), refc[4].copy(),
               refc[0][:65536].copy(), refc[0][:1024].copy(), refc[2][:4096 + k - 1].copy(),
               np.full(3000, 4, np.uint8),                                  # a contig of N only
               synth.mutate(rng, refc[0][10_000:30_000], 0.0, n_runs=1)]
    return spl, contigs


@pytest.mark.parametrize("k", [17, 21, 25, 31, 32])
def test_packed_scan_matches_oracle(hip_ctx, oracle, k):
    """pack -> scan on the 2-bit layout (contigs back to back at arbitrary symbol offsets, escaped blocks for N runs / IUPAC)
    must report what the oracle's scan reports"""
    import torch
    rng = np.random.default_rng(300 + k)
    spl, contigs = _packed_case(oracle, rng, k)
    off = np.zeros(len(contigs) + 1, np.uint64)
    off[1:] = np.cumsum([c.size for c in contigs])
    codes = np.concatenate(contigs)
    d = torch.from_numpy(codes).cuda()
    torch.cuda.synchronize()
    pk, keep = hip_ctx.pack_dev(d)
    hip_ctx.splitters_set(spl)
    got = hip_ctx.scan_packed_dev(pk, off, k)
    want = _oracle_hits(oracle, contigs, k, spl)
    for g, w, name in zip(got, want, ("ctg", "pos", "dir", "rc")):
        assert np.array_equal(g, w), name
    assert want[0].size > 100
    # and the expansion gives the symbols back, escaped blocks included
    out = torch.zeros(codes.size + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # (torch's fill and the library's kernels run on different streams)
    hip_ctx.expand_dev(pk, out.data_ptr())
    assert np.array_equal(out[:codes.size].cpu().numpy(), codes)


@pytest.mark.parametrize("k", [21, 31])
def test_packed_scan_with_a_saturated_suffix_filter(hip_ctx, oracle, k):
    """600 k decoy splitters that occur nowhere in the sample fill the 128 KiB suffix filter: most positions pass it, a wavefront has
    hundreds of survivors per step of 1024 positions -- the hand-out of survivors over the lanes takes several rounds -- and the
    second filter and the exact table must turn every one of them away: the hits are the oracle's"""
    import torch
    rng = np.random.default_rng(7000 + k)
    spl, contigs = _packed_case(oracle, rng, k)
    decoys = rng.integers(0, 1 << 62, 600_000, dtype=np.uint64) << np.uint64(2)
    decoys &= ~np.uint64((1 << (64 - 2 * k)) - 1)  # left-aligned k-mers
    all_spl = np.unique(np.concatenate([spl, decoys]))
    off = np.zeros(len(contigs) + 1, np.uint64)
    off[1:] = np.cumsum([c.size for c in contigs])
    d = torch.from_numpy(np.concatenate(contigs)).cuda()
    torch.cuda.synchronize()
    pk, keep = hip_ctx.pack_dev(d)
    hip_ctx.splitters_set(all_spl)
    got = hip_ctx.scan_packed_dev(pk, off, k)
    want = _oracle_hits(oracle, contigs, k, all_spl)
    for g, w, name in zip(got, want, ("ctg", "pos", "dir", "rc")):
        assert np.array_equal(g, w), name
    assert want[0].size > 100


@pytest.mark.parametrize("k", [21, 31])
def test_prefetched_scan_matches_oracle(hip_ctx, oracle, k):
    """agc_hip_prefetch_packed_dev + agc_hip_scan_prefetched (the next sample's scan queued ahead on its own stream) deliver
    what the oracle's scan reports; the packed sample reads back symbol for symbol (agc_hip_fetch_slices_packed); a
    scan_prefetched for another sample is refused"""
    import ctypes as C
    import torch
    from agc_amd import capi
    rng = np.random.default_rng(900 + k)
    hip_ctx.splitters_set(_packed_case(oracle, rng, k)[0])
    cases = []
    for _ in range(3):
        spl, contigs = _packed_case(oracle, rng, k)
        cases.append(contigs)
    hip_ctx.splitters_set(spl)
    packed = []
    for contigs in cases:
        off = np.zeros(len(contigs) + 1, np.uint64)
        off[1:] = np.cumsum([c.size for c in contigs])
        codes = np.concatenate(contigs)
        d = torch.from_numpy(codes).cuda()
        torch.cuda.synchronize()
        pk, keep = hip_ctx.pack_dev(d)
        packed.append((pk, keep, off, codes, contigs))
    for i, (pk, keep, off, codes, contigs) in enumerate(packed):
        hip_ctx.prefetch_packed_dev(pk, off, k)
        if i == 0:  # another sample than the one in flight
            other = packed[1]
            n = C.c_uint64()
            rc = hip_ctx.L.agc_hip_scan_prefetched(hip_ctx.h, C.byref(other[0]), capi._p(other[2], capi.u64p), other[2].size - 1, k, 0, C.byref(n), None, None, None, None)
            assert rc == capi.EINVAL
        got = hip_ctx.scan_prefetched(pk, off, k)
        want = _oracle_hits(oracle, contigs, k, spl)
        for g, w, name in zip(got, want, ("ctg", "pos", "dir", "rc")):
            assert np.array_equal(g, w), (i, name)
        back, _ = hip_ctx.fetch_slices_packed(pk, [0], [codes.size])
        assert np.array_equal(back, codes), i
        back, _ = hip_ctx.fetch_slices_packed(pk, [3], [codes.size - 5], rc=[1])
        assert np.array_equal(back, oracle.rev_comp(codes[3:codes.size - 2])), i


def test_packed_scan_equals_byte_scan_on_a_big_sample(hip_ctx, oracle):
    """size-independent property at a larger size: the two scan kernels agree (50 Mbp, 6 contigs, a few escaped blocks)"""
    import torch
    rng = np.random.default_rng(77)
    k = 31
    ref = synth.random_seq(rng, 2_000_000)
    spl = oracle.determine_splitters([ref], k, 20_000)
    parts = [synth.mutate(rng, ref, 0.001, n_runs=2, iupac=2) for _ in range(6)]
    parts += [np.tile(ref, 20)[: 38_000_000 - 7]]
    off = np.zeros(len(parts) + 1, np.uint64)
    off[1:] = np.cumsum([p.size for p in parts])
    d = torch.from_numpy(np.concatenate(parts)).cuda()
    torch.cuda.synchronize()
    pk, keep = hip_ctx.pack_dev(d)
    hip_ctx.splitters_set(spl)
    a = hip_ctx.scan_packed_dev(pk, off, k, cap=1 << 18)
    b = hip_ctx.scan_contigs_dev(d.data_ptr(), off, k, cap=1 << 18)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[0].size > 2000


# ---- segments and their groups on the device (agc_hip_segments_packed) ------------------------------------------------------
@pytest.mark.parametrize("k,prefetched,big_table", [(21, False, False), (31, True, False), (25, False, True)])
def test_segments_on_the_device_match_oracle(hip_ctx, oracle, k, prefetched, big_table):
    """hits -> reset rule -> cut -> key -> table look-up -> encode launch, all on the device, against the oracle's compress_contig
    per contig, a dictionary look-up of the keys, and the oracle's Encode of every segment whose group the table knows.  Both
    look-up variants (bucket ranges staged in LDS / straight from HBM: the table is made large against the batch), with and
    without the scan prefetched."""
    import torch
    rng = np.random.default_rng(4000 + k)
    spl, contigs = _packed_case(oracle, rng, k)
    off = np.zeros(len(contigs) + 1, np.uint64)
    off[1:] = np.cumsum([c.size for c in contigs])
    codes = np.concatenate(contigs)
    d = torch.from_numpy(codes).cuda()
    torch.cuda.synchronize()
    pk, keep = hip_ctx.pack_dev(d)
    hip_ctx.splitters_set(spl)
    # what the oracle cuts
    want = []
    for ci, c in enumerate(contigs):
        s = oracle.scan_contig(c, k, spl)
        for i in range(len(s["start"])):
            want.append((ci, int(s["start"][i]), int(s["len"][i]), int(s["front_dir"][i]), int(s["front_rc"][i]), int(s["front_full"][i]),
                         int(s["back_dir"][i]), int(s["back_rc"][i]), int(s["back_full"][i])))
    # groups: every other distinct key of the segments with two splitters gets a group whose reference is that segment, oriented
    # as the key says (add_segment, agc_compressor.cpp:1286-1301), and slightly edited so that the deltas are not empty
    mml, gid0 = 20, 20_000 + 1000 * k
    keys, refs = {}, {}
    for (ci, st, ln, fd, fr, ff, bd, br, bf) in want:
        if not (ff and bf):
            continue
        f, b = min(fd, fr), min(bd, br)
        key, rc = ((f, b), 0) if f < b else ((b, f), 1)
        if key in keys or (len(keys) + len(refs)) % 2 == 1:
            refs.setdefault(key, None)
            continue
        keys[key] = gid0 + len(keys)
        text = contigs[ci][st:st + ln]
        text = oracle.rev_comp(text) if rc else text
        ref = synth.mutate(rng, text, 0.003)
        hip_ctx.ref_register(keys[key], ref, mml)
        refs[key] = ref
    assert len(keys) > 20
    hip_ctx.group_map_set(keys, n_slots=(1 << 22) if big_table else 16)
    if prefetched:
        hip_ctx.prefetch_packed_dev(pk, off, k)
    segs, n_enc = hip_ctx.segments_packed(pk, off, k, prefetched=prefetched, encode_known=True, cap=16)
    assert len(segs) == len(want)
    n_known = 0
    texts = []
    for sg, (ci, st, ln, fd, fr, ff, bd, br, bf) in zip(segs, want):
        assert (int(sg["ctg"]), int(sg["start"]), int(sg["len"]), int(sg["front_full"]), int(sg["back_full"])) == (ci, st, ln, ff, bf)
        if ff:
            assert (int(sg["front_dir"]), int(sg["front_rc"])) == (fd, fr)
        if bf:
            assert (int(sg["back_dir"]), int(sg["back_rc"])) == (bd, br)
        if ff and bf:
            f, b = min(fd, fr), min(bd, br)
            key, rc = ((f, b), 0) if f < b else ((b, f), 1)
            assert int(sg["store_rc"]) == rc
            assert int(sg["map_gid"]) == keys.get(key, -1)
            if key in keys:
                assert sg["encoded"]
                n_known += 1
                t = contigs[ci][st:st + ln]
                texts.append((key, oracle.rev_comp(t) if rc else t))
            else:
                assert not sg["encoded"]
        else:
            assert int(sg["map_gid"]) == -1 and not sg["encoded"]
    assert n_enc == n_known and n_known > 20
    enc, eoff = hip_ctx.lz_encode_end()
    assert eoff.size == n_known + 1
    for i, (key, t) in enumerate(texts):
        assert np.array_equal(enc[int(eoff[i]):int(eoff[i + 1])], oracle.LZ(refs[key], mml).encode(t)), i
    # the launch as a call of its own (the table first, the encode when the caller's second lane is free): the same deltas
    segs3, _ = hip_ctx.segments_packed(pk, off, k, cap=1 << 14)
    m = hip_ctx.segments_encode_known(segs3, keys.values())
    assert int(m.sum()) == n_known
    enc3, eoff3 = hip_ctx.lz_encode_end()
    assert np.array_equal(enc3, enc) and np.array_equal(eoff3, eoff)
    # single slots of the table replaced: a key moves to another group, the look-up follows
    key0 = next(iter(keys))
    tab = hip_ctx.group_map_set(keys, n_slots=(1 << 22) if big_table else 16)
    idx = [i for i in range(tab.size) if tab[i]["used"] and (int(tab[i]["k1"]), int(tab[i]["k2"])) == key0]
    slot = tab[idx].copy()
    slot["gid"] = -7
    hip_ctx.group_map_update(idx, slot)
    segs2, _ = hip_ctx.segments_packed(pk, off, k, cap=1 << 14)
    moved = [int(sg["map_gid"]) for sg in segs2 if sg["front_full"] and sg["back_full"] and
             (min(int(sg["front_dir"]), int(sg["front_rc"])), min(int(sg["back_dir"]), int(sg["back_rc"]))) in (key0, key0[::-1])]
    assert moved and all(g == -7 for g in moved)


@pytest.mark.parametrize("seed", [9, 3, 14, 27])
def test_segments_of_fuzz_collections_match_oracle(hip_ctx, oracle, seed, tmp_path):
    """the cut on the device for the fuzzer's collections (tests/fuzz.py: tiny segment sizes -- hits closer than k --, contigs
    shorter than k, repeats, N runs): every file's segments against the oracle's compress_contig"""
    import torch
    from agc_amd import fasta
    from tests import fuzz
    case = fuzz.make_case(seed, str(tmp_path / "in"))
    k = int(case["args"][case["args"].index("-k") + 1])
    seg = int(case["args"][case["args"].index("-s") + 1])
    if k < 16:
        pytest.skip("the packed scan needs k >= 16")
    _n, rcodes, roff = fasta.read_codes(case["files"][0])
    spl = oracle.determine_splitters([rcodes[int(roff[i]):int(roff[i + 1])] for i in range(len(roff) - 1)], k, seg)
    hip_ctx.splitters_set(spl)
    hip_ctx.group_map_set({})
    for f in case["files"]:
        _names, codes, off = fasta.read_codes(f)
        if not codes.size:
            continue
        d = torch.from_numpy(np.concatenate([codes, np.zeros(64, np.uint8)])).cuda()
        torch.cuda.synchronize()
        pk, keep = hip_ctx.pack_dev(d, codes.size)
        segs, _ = hip_ctx.segments_packed(pk, off, k, cap=8)
        want = []
        for ci in range(len(off) - 1):
            s = oracle.scan_contig(codes[int(off[ci]):int(off[ci + 1])], k, spl)
            for i in range(len(s["start"])):
                want.append((ci, int(s["start"][i]), int(s["len"][i]), int(s["front_full"][i]), int(s["back_full"][i]),
                             int(s["front_dir"][i]) if s["front_full"][i] else 0, int(s["back_dir"][i]) if s["back_full"][i] else 0))
        got = [(int(g["ctg"]), int(g["start"]), int(g["len"]), int(g["front_full"]), int(g["back_full"]), int(g["front_dir"]), int(g["back_dir"])) for g in segs]
        assert got == want, (f, len(got), len(want), next((i, a, b) for i, (a, b) in enumerate(zip(got + [None], want + [None])) if a != b))
