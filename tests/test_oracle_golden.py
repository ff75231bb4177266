"""CPU: the oracle (oracle/agc_oracle.c) against the golden vectors generated from the reference
itself (tests/golden/make_golden.py), and -- where oracle/_ref exists -- against the reference live."""
import os

import numpy as np
import pytest

from agc_amd import synth
from tests.cases import lz_cases

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lz_golden(oracle):
    d = np.load(os.path.join(G, "lz_golden.npz"))
    n = int(d["n"][0])
    assert n >= 40
    for i in range(n):
        mml = int(d[f"mml_{i}"][0])
        ref, text = d[f"ref_{i}"], d[f"text_{i}"]
        z = oracle.LZ(ref, mml)
        assert np.array_equal(z.encode(text), d[f"enc_{i}"]), f"encode {i}"
        bounds = [0xFFFFFFFF, 0, 10, 100, text.size // 2]
        got = [z.estimate(text, b) for b in bounds]
        assert got == [int(x) for x in d[f"est_{i}"]], f"estimate {i}"
        assert np.array_equal(z.cost_vector(text, 0), d[f"cv0_{i}"]), f"cost vector suffix {i}"
        assert np.array_equal(z.cost_vector(text, 1), d[f"cv1_{i}"]), f"cost vector prefix {i}"
        enc = d[f"enc_{i}"]
        if enc.size:
            dec, m = z.decode(enc, text.size + 8)
            assert m == text.size and np.array_equal(dec, text), f"decode {i}"


def test_scan_golden(oracle):
    d = np.load(os.path.join(G, "scan_golden.npz"))
    for i in range(int(d["n"][0])):
        k = int(d[f"k_{i}"][0])
        s = oracle.scan_contig(d[f"ctg_{i}"], k, d[f"spl_{i}"])
        m = s["back_full"] == 1
        assert np.array_equal(s["start"][m] + s["len"][m] - 1, d[f"pos_{i}"])
        assert np.array_equal(s["back_dir"][m], d[f"dir_{i}"])
        assert np.array_equal(s["back_rc"][m], d[f"rc_{i}"])
        # segments tile the contig with k-symbol overlaps
        assert s["start"][0] == 0 and int(s["start"][-1] + s["len"][-1]) == d[f"ctg_{i}"].size
        assert np.all(s["start"][1:] == s["start"][:-1] + s["len"][:-1] - k)


def test_estimate_peak_replays_bound(oracle):
    for mml, ref, text in lz_cases(seed=5, n_cases=24):
        z = oracle.LZ(ref, mml)
        full, peak = z.estimate(text, want_peak=True)
        for bound in (0, 7, 50, 400, text.size):
            b = z.estimate(text, bound)
            if peak > bound:
                assert b > bound
            else:
                assert b == full


def test_tuples_and_repetitiveness(oracle):
    rng = np.random.default_rng(1)
    a = synth.random_seq(rng, 1001)
    t = oracle.bytes2tuples(a)
    assert t.size == 1001 // 4 + 1 + 1 and t[-1] == (4 << 4) + 1
    b = a.copy(); b[10] = 4
    t = oracle.bytes2tuples(b)
    assert t[-1] == (3 << 4) + (1001 % 3)
    c = a.copy(); c[10] = 15
    assert oracle.bytes2tuples(c)[-1] == (2 << 4) + 1
    e = a.copy(); e[10] = 30
    t = oracle.bytes2tuples(e)
    assert t.size == 1002 and t[-1] == 0x10
    assert not oracle.ref_is_repetitive(a)
    assert oracle.ref_is_repetitive(np.tile(synth.random_seq(rng, 6), 300))
    assert oracle.bytes2tuples(np.zeros(0, np.uint8)).size == 2


def test_preprocess_table(oracle):
    raw = np.frombuffer(b"ACGTacgtNnRYSWKMBDHVUXZ@`\n\r 09>", np.uint8)
    got = oracle.preprocess(raw)
    assert list(got) == [0, 1, 2, 3, 0, 1, 2, 3, 4, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 30, 30, 32, 32]
    assert np.array_equal(oracle.rev_comp(np.array([0, 1, 2, 3, 4, 30], np.uint8)), np.array([30, 4, 0, 1, 2, 3], np.uint8))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "libagcref.so")),
                    reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_reference_live(oracle):
    """fresh seeds every case family, compared with the reference's own lz_diff.cpp compiled in place"""
    for i, (mml, ref, text) in enumerate(lz_cases(seed=99, n_cases=40)):
        a, b = oracle.LZ(ref, mml), oracle.RefLZ(ref, mml)
        assert np.array_equal(a.encode(text), b.encode(text)), i
        assert a.estimate(text) == b.estimate(text), i
        assert a.estimate(text, 25) == b.estimate(text, 25), i
        for pc in (0, 1):
            assert np.array_equal(a.cost_vector(text, pc), b.cost_vector(text, pc)), i


def test_container_reader_on_reference_archive():
    """the .agc container reader used by the parity tests, on the archive the reference wrote for
    BASELINE configs[0] (toy_ex, -k 25 -l 17): stream order and part framing of SURVEY App. A.8 / B.2"""
    import hashlib
    import json
    from tests import agc_container
    data = open(os.path.join(G, "toy_c1_reference.agc"), "rb").read()
    gold = json.load(open(os.path.join(G, "archives.json")))["toy_c1"]
    assert hashlib.sha256(data).hexdigest() == gold["sha256"] == "81502256be60722f89f07622e55a222bce43f558f8a0d760d73313acdb977d62"
    streams, order = agc_container.parse(data)
    assert order[:3] == ["collection-samples", "collection-contigs", "collection-details"]
    assert order[3:19] == ["x%sd" % c for c in "0123456789ABCDEF"]
    assert order[-4:] == ["params", "splitters", "segment-splitters", "file_type_info"]
    # contigs shorter than k: no splitters, every contig is a raw segment; 16 raw groups pre-seeded with 0x7f
    assert streams["splitters"][0][1] == b""
    import struct
    assert struct.unpack("<4I", streams["params"][0][1]) == (25, 17, 50, 60000)
    assert sum(len(p) for n, p in streams.items() if n.startswith("x") and n.endswith("d")) >= 16
    assert agc_container.diff(data, data) == []


def test_lz_properties_on_random_edits(oracle):
    """size-independent properties of the LZ-diff restatement, on random references and edit mixes (hypothesis):
    decode(encode(t)) == t; the identical sequence encodes to nothing; the unbounded estimate (value, peak) is reproducible;
    both cost vectors cover every position with a positive total."""
    from hypothesis import given, settings, strategies as hs

    @settings(max_examples=60, deadline=None)
    @given(seed=hs.integers(0, 2**31 - 1), n=hs.integers(1, 4000), mml=hs.integers(15, 32),
           d=hs.sampled_from([0.0, 0.001, 0.02, 0.2]), n_runs=hs.integers(0, 3), iupac=hs.integers(0, 3), indels=hs.integers(0, 3))
    def prop(seed, n, mml, d, n_runs, iupac, indels):
        rng = np.random.default_rng(seed)
        ref = synth.random_seq(rng, n)
        if rng.random() < 0.2:
            ref[int(rng.integers(0, n))] = 4
        text = synth.mutate(rng, ref, d, n_runs=n_runs, iupac=iupac, indels=indels)
        if text.size == 0:
            return
        z = oracle.LZ(ref, mml)
        enc = z.encode(text)
        if np.array_equal(text, ref):
            assert enc.size == 0
        else:
            dec, m = z.decode(enc, text.size + 8)
            assert m == text.size and np.array_equal(dec, text)
        c1, p1 = z.estimate(text, want_peak=True)
        c2, p2 = z.estimate(text, want_peak=True)
        assert (c1, p1) == (c2, p2) and p1 <= 0xFFFFFFFF
        for pf in (0, 1):
            cv = z.cost_vector(text, pf)
            assert cv.size == text.size and int(cv.sum()) > 0

    prop()
