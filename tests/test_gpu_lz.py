"""GPU parity: HIP LZ-diff kernels vs the oracle, through the C ABI (bit-exact)."""
import numpy as np
import pytest

from agc_amd import synth
from tests.cases import lz_cases

pytestmark = pytest.mark.gpu


def _register(hip_ctx, cases, gid0):
    for i, (mml, ref, _t) in enumerate(cases):
        hip_ctx.ref_register(gid0 + i, ref, mml)


@pytest.fixture(scope="module")
def registered(hip_ctx):
    cases = lz_cases()
    # references with different min_match_len are independent groups
    _register(hip_ctx, cases, 1000)
    return cases


def test_index_matches_oracle(hip_ctx, oracle, registered):
    for i, (mml, ref, _t) in enumerate(registered):
        tab, is16 = hip_ctx.ref_index_get(1000 + i)
        want = oracle.LZ(ref, mml).index()
        assert is16 == (want.dtype == np.uint16)
        assert np.array_equal(tab.astype(np.uint32), want.astype(np.uint32)), f"case {i}"
        assert np.array_equal(hip_ctx.ref_get(1000 + i), ref)


def _concat(cases):
    texts = [t for (_m, _r, t) in cases]
    off = np.zeros(len(texts), np.uint64)
    ln = np.array([t.size for t in texts], np.uint32)
    off[1:] = np.cumsum(ln[:-1].astype(np.uint64) + 7)  # ragged gaps between texts
    buf = np.full(int(off[-1] + ln[-1]) + 8, 4, np.uint8)
    for o, t in zip(off, texts):
        buf[int(o):int(o) + t.size] = t
    return buf, off, ln


def test_encode_bit_exact(hip_ctx, oracle, registered):
    buf, off, ln = _concat(registered)
    gids = 1000 + np.arange(len(registered))
    enc, eoff = hip_ctx.lz_encode_batch(buf, gids, off, ln)
    for i, (mml, ref, text) in enumerate(registered):
        want = oracle.LZ(ref, mml).encode(text)
        got = enc[int(eoff[i]):int(eoff[i + 1])]
        assert np.array_equal(got, want), f"case {i} mml={mml} len={text.size}: got {got[:40].tobytes()} want {want[:40].tobytes()}"
        if want.size:
            dec, n = oracle.LZ(ref, mml).decode(got, text.size + 8)
            assert n == text.size and np.array_equal(dec, text)


def test_encode_reverse_complement(hip_ctx, oracle, registered):
    # rc flag: the kernel must see reverse_complement_copy(text)
    buf, off, ln = _concat(registered)
    gids = 1000 + np.arange(len(registered))
    rc = np.ones(len(registered), np.uint8)
    enc, eoff = hip_ctx.lz_encode_batch(buf, gids, off, ln, rc=rc)
    for i, (mml, ref, text) in enumerate(registered):
        want = oracle.LZ(ref, mml).encode(oracle.rev_comp(text))
        assert np.array_equal(enc[int(eoff[i]):int(eoff[i + 1])], want), f"case {i}"


def test_encode_in_two_halves_beside_other_calls(hip_ctx, oracle, registered):
    """agc_hip_lz_encode_begin_dev / _end (second stream, own scratch): estimates and cost vectors issued between the two
    halves come out right, and the deltas are the ones the one-call encode and the oracle give -- also for reverse-complemented
    texts and after a too-small buffer (AGC_HIP_ECAP, second end call)"""
    import torch
    buf, off, ln = _concat(registered)
    gids = 1000 + np.arange(len(registered))
    rc = (np.arange(len(registered)) % 2).astype(np.uint8)
    d = torch.from_numpy(np.concatenate([buf, np.zeros(64, np.uint8)])).cuda()
    torch.cuda.synchronize()
    for round_ in range(3):
        hip_ctx.lz_encode_begin_dev(d.data_ptr(), gids, off, ln, rc=rc)
        cost, peak = hip_ctx.lz_estimate_batch_dev(d.data_ptr(), gids, off, ln)
        pf = np.zeros(len(registered), np.uint8)
        costs = hip_ctx.lz_cost_vector_batch_dev(d.data_ptr(), gids, off, ln, None, pf)
        enc, eoff = hip_ctx.lz_encode_end(enc_cap=16 if round_ == 1 else None)
        p = 0
        for i, (mml, ref, text) in enumerate(registered):
            z = oracle.LZ(ref, mml)
            want = z.encode(oracle.rev_comp(text) if rc[i] else text)
            assert np.array_equal(enc[int(eoff[i]):int(eoff[i + 1])], want), f"round {round_} case {i}"
            assert int(cost[i]) == z.estimate(text), f"round {round_} case {i}"
            assert np.array_equal(costs[p:p + text.size], z.cost_vector(text, 0)), f"round {round_} case {i}"
            p += text.size
    one, ooff = hip_ctx.lz_encode_batch_dev(d.data_ptr(), gids, off, ln, rc=rc)
    assert np.array_equal(one, enc) and np.array_equal(ooff, eoff)


def test_estimate_and_peak(hip_ctx, oracle, registered):
    buf, off, ln = _concat(registered)
    gids = 1000 + np.arange(len(registered))
    cost, peak = hip_ctx.lz_estimate_batch(buf, gids, off, ln)
    for i, (mml, ref, text) in enumerate(registered):
        z = oracle.LZ(ref, mml)
        want, wpeak = z.estimate(text, want_peak=True)
        assert int(cost[i]) == want, f"case {i}"
        assert int(peak[i]) == wpeak, f"case {i}"
        # replaying a bound from (cost, peak) reproduces the bounded call's accept/reject
        for bound in (0, 10, 100, text.size // 2):
            bounded = z.estimate(text, bound)
            if wpeak > bound:
                assert bounded > bound
            else:
                assert bounded == want


def test_cost_vector(hip_ctx, oracle, registered):
    buf, off, ln = _concat(registered)
    gids = 1000 + np.arange(len(registered))
    for prefix in (0, 1):
        pf = np.full(len(registered), prefix, np.uint8)
        costs = hip_ctx.lz_cost_vector_batch(buf, gids, off, ln, None, pf)
        p = 0
        for i, (mml, ref, text) in enumerate(registered):
            want = oracle.LZ(ref, mml).cost_vector(text, prefix)
            assert np.array_equal(costs[p:p + text.size], want), f"case {i} prefix={prefix}"
            p += text.size


def test_long_reference_u32_table(hip_ctx, oracle):
    rng = np.random.default_rng(11)
    from agc_amd import synth
    ref = synth.random_seq(rng, 270_000)          # ref/4 >= 65535 -> 32-bit table (lz_diff.cpp:146)
    text = synth.mutate(rng, ref, 0.003, n_runs=2, indels=2)
    hip_ctx.ref_register(5000, ref, 20)
    tab, is16 = hip_ctx.ref_index_get(5000)
    z = oracle.LZ(ref, 20)
    assert not is16 and np.array_equal(tab, z.index())
    enc, eoff = hip_ctx.lz_encode_batch(text, [5000], [0], [text.size])
    assert np.array_equal(enc, z.encode(text))
    cost, _ = hip_ctx.lz_estimate_batch(text, [5000], [0], [text.size])
    assert int(cost[0]) == z.estimate(text)


def test_unregistered_group_is_an_error(hip_ctx):
    from agc_amd import capi
    with pytest.raises(capi.AgcHipError) as e:
        hip_ctx.lz_encode_batch(np.zeros(100, np.uint8), [999999], [0], [100])
    assert e.value.code == capi.ENOREF


def test_lag_counts(hip_ctx, oracle):
    import torch
    rng = np.random.default_rng(3)
    from agc_amd import synth
    seqs = [synth.random_seq(rng, 5000), np.tile(synth.random_seq(rng, 5), 1000), synth.mutate(rng, synth.random_seq(rng, 3000), 0, n_runs=4)]
    buf = np.concatenate(seqs)
    off = np.cumsum([0] + [s.size for s in seqs[:-1]]).astype(np.uint64)
    ln = np.array([s.size for s in seqs], np.uint32)
    d = torch.from_numpy(buf).cuda()
    for rc in (0, 1):
        cnt, cur = hip_ctx.ref_lag_counts_dev(d.data_ptr(), off, ln, np.full(3, rc, np.uint8))
        for i, s in enumerate(seqs):
            wc, wu = oracle.ref_lag_counts(oracle.rev_comp(s) if rc else s)
            assert np.array_equal(cnt[i], wc) and np.array_equal(cur[i], wu)


def _split_point_oracle(oracle, ref1, ref2, mml, seg, rc1, pf1, rc2, pf2):
    """find_cand_segment_with_missing_middle_splitter, src/core/agc_compressor.cpp:1540-1625 (before the k+1 clamps):
    v1 = prefix sums of the first cost vector (reversed first when it was taken with prefix_costs = false),
    v2 = suffix sums of the second one (taken on the reversed vector when prefix_costs = true); first arg-min of v1 + v2,
    all in u32 as std::partial_sum on vector<uint32_t>."""
    t1 = oracle.rev_comp(seg) if rc1 else seg
    t2 = oracle.rev_comp(seg) if rc2 else seg
    c1 = oracle.LZ(ref1, mml).cost_vector(t1, int(pf1)).astype(np.uint32)
    c2 = oracle.LZ(ref2, mml).cost_vector(t2, int(pf2)).astype(np.uint32)
    if not pf1:
        c1 = c1[::-1]
    v1 = np.cumsum(c1, dtype=np.uint32)
    if pf2:
        v2 = np.cumsum(c2, dtype=np.uint32)[::-1]
    else:
        v2 = np.cumsum(c2[::-1], dtype=np.uint32)[::-1]
    s = (v1 + v2).astype(np.uint32)
    p = int(np.argmin(s))  # first minimum
    return p, int(s[p])


def test_split_point_matches_oracle(hip_ctx, oracle):
    """agc_hip_lz_split_point_batch_dev against the oracle's two cost vectors + the reference's sums / arg-min, for every
    orientation combination; segments = left part of one reference + right part of another (the destroyed-middle-splitter
    shape), plus unrelated and tiny segments."""
    import torch
    from agc_amd import synth
    rng = np.random.default_rng(4242)
    mml = 18
    k = 21
    gid0 = 7000
    refs, segs = [], []
    for case in range(12):
        n1, n2 = int(rng.integers(300, 6000)), int(rng.integers(300, 6000))
        a, b = synth.random_seq(rng, n1), synth.random_seq(rng, n2)
        b[:k] = a[-k:]  # the two references overlap by the middle splitter
        if case == 9:
            seg = synth.random_seq(rng, 500)  # nothing matches: all literals, arg-min at the first position
        elif case == 10:
            seg = a[:30].copy()
        else:
            seg = np.concatenate([a, b[k:]])
            seg = synth.mutate(rng, seg, 0.01, n_runs=1 if case % 3 == 0 else 0, indels=case % 2)
        refs += [a, b]
        segs.append(seg)
    for i, r in enumerate(refs):
        hip_ctx.ref_register(gid0 + i, r, mml)
    off = np.zeros(len(segs), np.uint64)
    ln = np.array([s.size for s in segs], np.uint32)
    off[1:] = np.cumsum(ln[:-1].astype(np.uint64) + 5)
    buf = np.full(int(off[-1] + ln[-1]) + 64, 4, np.uint8)
    for o, s in zip(off, segs):
        buf[int(o):int(o) + s.size] = s
    d = torch.from_numpy(buf).cuda()
    g1, g2, oo, ll, r1, p1, r2, p2, want = [], [], [], [], [], [], [], [], []
    for i, seg in enumerate(segs):
        for combo in range(16):
            rc1, pf1, rc2, pf2 = combo & 1, (combo >> 1) & 1, (combo >> 2) & 1, (combo >> 3) & 1
            # the reference only uses (rc1 == !pf1) xor use_rc combinations, but the ABI takes any
            g1.append(gid0 + 2 * i), g2.append(gid0 + 2 * i + 1), oo.append(off[i]), ll.append(ln[i])
            r1.append(rc1), p1.append(pf1), r2.append(rc2), p2.append(pf2)
            want.append(_split_point_oracle(oracle, refs[2 * i], refs[2 * i + 1], mml, seg, rc1, pf1, rc2, pf2))
    pos, sm = hip_ctx.lz_split_point_batch_dev(d.data_ptr(), g1, g2, oo, ll, r1, p1, r2, p2)
    for j, (wp, ws) in enumerate(want):
        assert (int(pos[j]), int(sm[j])) == (wp, ws), f"job {j} (segment {j // 16}, combo {j % 16}): got {(int(pos[j]), int(sm[j]))} want {(wp, ws)}"


# ---------------------------------------------------------------------------------------------------------------------
# the same entry points on sequences of a 2-bit packed sample (agc_hip_*_packed): what the create path calls.  The byte-input
# tests above reach the same kernels through a packed copy whose blocks are nearly all escaped (their texts sit between N
# gaps); here the texts lie in clean blocks, at every alignment inside the 16-symbol words, so that the 64-bit XOR compare,
# the funnel shifts and the reversed reads of reverse-complemented texts are what runs.
# ---------------------------------------------------------------------------------------------------------------------
def _packed_sample(hip_ctx, texts, rng, gap_fill="acgt"):
    """texts back to back with ragged gaps of random ACGT (clean blocks) or of N (escaped blocks) -> (Packed, keep, off, len, buf)"""
    import torch
    off, parts, o = [], [], 0
    for i, t in enumerate(texts):
        g = int(rng.integers(1, 40))
        parts.append(rng.integers(0, 4, g).astype(np.uint8) if gap_fill == "acgt" else np.full(g, 4, np.uint8))
        o += g
        off.append(o)
        parts.append(t)
        o += t.size
    parts.append(rng.integers(0, 4, 70).astype(np.uint8))
    buf = np.concatenate(parts)
    d = torch.from_numpy(buf).cuda()
    torch.cuda.synchronize()
    pk, keep = hip_ctx.pack_dev(d)
    return pk, (keep, d), np.array(off, np.uint64), np.array([t.size for t in texts], np.uint32), buf


@pytest.mark.parametrize("gap_fill", ["acgt", "n"])
def test_packed_entry_points_match_oracle(hip_ctx, oracle, registered, gap_fill):
    rng = np.random.default_rng(2024)
    texts = [t for (_m, _r, t) in registered]
    pk, keep, off, ln, buf = _packed_sample(hip_ctx, texts, rng, gap_fill)
    gids = 1000 + np.arange(len(registered))
    for rcflag in (0, 1):
        rc = np.full(len(registered), rcflag, np.uint8)
        enc, eoff = hip_ctx.lz_encode_batch_packed(pk, gids, off, ln, rc=rc)
        cost, peak = hip_ctx.lz_estimate_batch_packed(pk, gids, off, ln, rc=rc)
        for prefix in (0, 1):
            costs = hip_ctx.lz_cost_vector_batch_packed(pk, gids, off, ln, rc, np.full(len(registered), prefix, np.uint8))
            p = 0
            for i, (mml, ref, text) in enumerate(registered):
                t = oracle.rev_comp(text) if rcflag else text
                assert np.array_equal(costs[p:p + t.size], oracle.LZ(ref, mml).cost_vector(t, prefix)), f"cost vector case {i} rc={rcflag} prefix={prefix}"
                p += t.size
        for i, (mml, ref, text) in enumerate(registered):
            z = oracle.LZ(ref, mml)
            t = oracle.rev_comp(text) if rcflag else text
            assert np.array_equal(enc[int(eoff[i]):int(eoff[i + 1])], z.encode(t)), f"encode case {i} rc={rcflag} gaps={gap_fill}"
            want, wpeak = z.estimate(t, want_peak=True)
            assert (int(cost[i]), int(peak[i])) == (want, wpeak), f"estimate case {i} rc={rcflag}"
    # the encode in two halves, the other entry points in between
    hip_ctx.lz_encode_begin_packed(pk, gids, off, ln)
    back, boff = hip_ctx.fetch_slices_packed(pk, off, ln, rc=(np.arange(len(registered)) % 2).astype(np.uint8))
    enc2, eoff2 = hip_ctx.lz_encode_end()
    one, ooff = hip_ctx.lz_encode_batch_packed(pk, gids, off, ln)
    assert np.array_equal(enc2, one) and np.array_equal(eoff2, ooff)
    # two encodes in flight, one per lane (AGC_HIP_ENCODE_LANES), collected in the other order; lane 1 with every other segment
    half = np.arange(0, len(registered), 2)
    hip_ctx.lz_encode_begin_packed(pk, gids, off, ln, lane=0)
    hip_ctx.lz_encode_begin_packed(pk, gids[half], off[half], ln[half], rc=np.ones(half.size, np.uint8), lane=1)
    cost2, _ = hip_ctx.lz_estimate_batch_packed(pk, gids, off, ln)
    enc_b, eoff_b = hip_ctx.lz_encode_end_on(1)
    enc_a, eoff_a = hip_ctx.lz_encode_end_on(0)
    assert np.array_equal(enc_a, one) and np.array_equal(eoff_a, ooff)
    want_b, woff_b = hip_ctx.lz_encode_batch_packed(pk, gids[half], off[half], ln[half], rc=np.ones(half.size, np.uint8))
    assert np.array_equal(enc_b, want_b) and np.array_equal(eoff_b, woff_b)
    for i, (_m, _r, text) in enumerate(registered):
        assert np.array_equal(back[int(boff[i]):int(boff[i + 1])], oracle.rev_comp(text) if i % 2 else text), i
    cnt, cur = hip_ctx.ref_lag_counts_packed(pk, off, ln)
    for i, (_m, _r, text) in enumerate(registered):
        wc, wu = oracle.ref_lag_counts(text)
        assert np.array_equal(cnt[i], wc) and np.array_equal(cur[i], wu), i


def test_packed_references_out_of_a_sample(hip_ctx, oracle):
    """agc_hip_ref_register_batch_packed: new references cut out of a packed sample at odd offsets, both orientations, clean and
    with N runs / IUPAC codes (escape blocks of the stored form): stored symbols, index tables, and encodes against them"""
    from agc_amd import synth
    rng = np.random.default_rng(77)
    mml = 20
    refs = [synth.random_seq(rng, n) for n in (37, 1000, 1024, 5000, 70_001)]
    refs += [synth.mutate(rng, synth.random_seq(rng, 9000), 0.0, n_runs=3, iupac=3), synth.mutate(rng, synth.random_seq(rng, 2500), 0.0, n_runs=1)]
    pk, keep, off, ln, buf = _packed_sample(hip_ctx, refs, rng, "acgt")
    rc = (np.arange(len(refs)) % 2).astype(np.uint8)
    gid0 = 9000
    hip_ctx.ref_register_batch_packed(gid0 + np.arange(len(refs)), pk, off, ln, rc, mml)
    texts = []
    for i, r in enumerate(refs):
        stored = oracle.rev_comp(r) if rc[i] else r
        assert np.array_equal(hip_ctx.ref_get(gid0 + i), stored), i
        tab, is16 = hip_ctx.ref_index_get(gid0 + i)
        want = oracle.LZ(stored, mml).index()
        assert np.array_equal(tab.astype(np.uint32), want.astype(np.uint32)), i
        texts.append(synth.mutate(rng, stored, 0.004, n_runs=1 if i % 2 else 0, indels=1) if stored.size > 200 else stored.copy())
    pk2, keep2, off2, ln2, _ = _packed_sample(hip_ctx, texts, rng, "acgt")
    enc, eoff = hip_ctx.lz_encode_batch_packed(pk2, gid0 + np.arange(len(refs)), off2, ln2)
    for i, r in enumerate(refs):
        stored = oracle.rev_comp(r) if rc[i] else r
        assert np.array_equal(enc[int(eoff[i]):int(eoff[i + 1])], oracle.LZ(stored, mml).encode(texts[i])), i


# ---- the parse in chunks (lz_kernels.hip: ChunkCtl; launch_parse picks it for launches of few, long texts) --------------------------
def _long_cases(oracle):
    """few, long, diverged texts -- the launches the chunked parse is made for: 2.5 % substitutions with indels, a text whose second
    half belongs to another reference (the shape of a missing-middle candidate), N runs, a text identical to its reference, a
    low-complexity reference, a text with a novel insertion of 30 kb (no match end for several chunks)"""
    rng = np.random.default_rng(515)
    out = []
    ra, rb = synth.random_seq(rng, 260_000), synth.random_seq(rng, 240_000)
    out.append((20, ra, synth.mutate(rng, ra, 0.025, indels=12)))
    out.append((15, rb, synth.mutate(rng, rb, 0.05, n_runs=6, iupac=5, indels=6)))
    half = np.concatenate([synth.mutate(rng, ra[:130_000], 0.025), synth.mutate(rng, rb[100_000:220_000], 0.025)])
    out.append((20, ra, half))
    out.append((20, rb, half))
    out.append((20, ra, ra.copy()))
    unit = synth.random_seq(rng, 7)
    low = np.tile(unit, 20_000)[:120_000].copy()
    out.append((17, low, synth.mutate(rng, low, 0.01)))
    ins = synth.mutate(rng, ra[:200_000], 0.01)
    out.append((20, ra, np.concatenate([ins[:90_000], synth.random_seq(rng, 30_000), ins[90_000:]])))
    big = synth.mutate(rng, rb, 0.002)
    big[50_000:58_000] = 4
    out.append((24, rb, big))
    return out


def test_chunked_parse_of_few_long_texts(hip_ctx, oracle):
    """encode, estimate + peak and both cost vectors of launches the library parses in chunks (a wavefront per 4096-symbol chunk,
    then one per text that joins them at the match ends both parses share) against the oracle's sequential parse"""
    cases = _long_cases(oracle)
    for i, (mml, ref, _t) in enumerate(cases):
        hip_ctx.ref_register(3000 + i, ref, mml)
    buf, off, ln = _concat(cases)
    gids = 3000 + np.arange(len(cases))
    for rc in (None, np.ones(len(cases), np.uint8)):
        enc, eoff = hip_ctx.lz_encode_batch(buf, gids, off, ln, rc=rc)
        for i, (mml, ref, text) in enumerate(cases):
            t = text if rc is None else oracle.rev_comp(text)
            want = oracle.LZ(ref, mml).encode(t)
            got = enc[int(eoff[i]):int(eoff[i + 1])]
            if not np.array_equal(got, want):
                d = int(np.argmax(got[:min(got.size, want.size)] != want[:min(got.size, want.size)])) if min(got.size, want.size) else 0
                raise AssertionError(f"case {i} rc={rc is not None}: {got.size} / {want.size} bytes, first difference at {d}: "
                                     f"{got[max(d - 20, 0):d + 20].tobytes()} / {want[max(d - 20, 0):d + 20].tobytes()}")
    cost, peak = hip_ctx.lz_estimate_batch(buf, gids, off, ln)
    for i, (mml, ref, text) in enumerate(cases):
        want, wpeak = oracle.LZ(ref, mml).estimate(text, want_peak=True)
        assert (int(cost[i]), int(peak[i])) == (want, wpeak), f"case {i}"
    for prefix in (0, 1):
        costs = hip_ctx.lz_cost_vector_batch(buf, gids, off, ln, None, np.full(len(cases), prefix, np.uint8))
        p = 0
        for i, (mml, ref, text) in enumerate(cases):
            want = oracle.LZ(ref, mml).cost_vector(text, prefix)
            got = costs[p:p + text.size]
            if not np.array_equal(got, want):
                d = int(np.argmax(got != want))
                raise AssertionError(f"case {i} prefix={prefix}: first difference at {d}: {got[max(d - 8, 0):d + 8]} / {want[max(d - 8, 0):d + 8]}")
            p += text.size


@pytest.mark.parametrize("chunk", [64, 300, 1024])
def test_lz_suite_with_every_launch_parsed_in_chunks(chunk):
    """AGC_HIP_LZ_CHUNK=<symbols> forces the chunked parse onto every launch the host holds descriptors of: this file's other tests
    and the archive tests (one-splitter estimates, missing-middle cost vectors, encodes of every collection) run again that way,
    with chunks from 64 symbols (a chunk boundary every few tokens) to 1024"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AGC_HIP_LZ_CHUNK=str(chunk))
    sel = ["tests/test_gpu_lz.py", "-k", "not every_launch_parsed_in_chunks"]
    if chunk == 300:
        sel = ["tests/test_gpu_lz.py", "tests/test_gpu_archive.py", "-k", "not every_launch_parsed_in_chunks and not full_size"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q"] + sel, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:]
