"""Device-side synthetic collections for bench.py (SURVEY.md §8d generator, torch on HBM).

A collection = one reference (contig lengths proportional to GRCh38 chr1-22,X,Y, uniform
i.i.d. ACGT) and samples = reference with independent per-base substitutions at rate d.
Everything is generated directly in HBM as one byte per symbol (codes 0..3).

Splitters: bench.py runs the real determine_splitters on the GPU (agc_hip_determine_splitters_dev);
positional_splitters() below is the optional shortcut (--positional-splitters).  The reference picks, in every reference contig, the first SINGLETON k-mer seen
once >= segment_size symbols have passed since the previous splitter, plus the right-most
singleton of the tail (src/core/agc_compressor.cpp:762-825).  For an i.i.d. random
reference of <= a few Gbp and k >= 25 essentially every k-mer is a singleton
(expected colliding pairs ~ n^2 / 4^k), so the generator applies the positional rule
directly; the once-per-archive singleton determination (sorting all reference k-mers,
agc_compressor.cpp:428-563) is reference preprocessing, not the per-sample hot path.
"""
import numpy as np
import torch

# GRCh38 primary chromosome lengths (Mbp, rounded) -- only the proportions matter
GRCH38_MBP = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]


def contig_lengths(total_bp, n_contigs=24):
    w = np.array(GRCH38_MBP[:n_contigs], dtype=np.float64)
    ln = np.maximum((w / w.sum() * total_bp).astype(np.int64), 1000)
    return ln


def make_reference(total_bp, seed, device, n_contigs=24):
    ln = contig_lengths(total_bp, n_contigs)
    off = np.zeros(n_contigs + 1, np.uint64)
    off[1:] = np.cumsum(ln)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    tot = int(off[-1])
    ref = torch.empty(tot + 4096, dtype=torch.uint8, device=device)
    step = 1 << 28
    for b in range(0, tot, step):
        e = min(tot, b + step)
        ref[b:e] = torch.randint(0, 4, (e - b,), dtype=torch.uint8, device=device, generator=g)
    ref[tot:] = 4
    return ref, off


def make_sample(ref, total, d, seed, device):
    """substitution at rate d, new base uniform among the other three"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = ref.clone()
    step = 1 << 27
    for b in range(0, total, step):
        e = min(total, b + step)
        m = torch.rand(e - b, device=device, generator=g) < d
        idx = m.nonzero(as_tuple=True)[0]
        delta = torch.randint(1, 4, (idx.numel(),), dtype=torch.uint8, device=device, generator=g)
        out[b + idx] = (ref[b + idx] + delta) & 3
    return out


def kmers_ending_at(seq, ends, k):
    """canonical k-mers (left-aligned u64 as CKmer keeps them) ending at absolute positions `ends` (torch int64)."""
    dev = seq.device
    idx = ends.unsqueeze(1) - (k - 1) + torch.arange(k, device=dev).unsqueeze(0)
    s = seq[idx].to(torch.int64)                                   # n x k, first symbol first
    sh = torch.arange(k - 1, -1, -1, device=dev, dtype=torch.int64) * 2
    d = (s << sh).sum(dim=1)                                       # right-aligned; k == 32 wraps into the sign bit, fine
    rc = ((3 - s) << (torch.arange(0, k, device=dev, dtype=torch.int64) * 2)).sum(dim=1)
    d = d.cpu().numpy().astype(np.uint64)
    rc = rc.cpu().numpy().astype(np.uint64)
    sl = np.uint64(64 - 2 * k)
    d, rc = d << sl, rc << sl
    return np.minimum(d, rc), d, rc


def positional_splitters(ref, ctg_off, k, segment_size):
    """ends of the splitter k-mers in every reference contig + their canonical values"""
    ends = []
    for c in range(len(ctg_off) - 1):
        b, e = int(ctg_off[c]), int(ctg_off[c + 1])
        if e - b < k:
            continue
        # first k-mer of the contig, then every first k-mer whose running length reached segment_size
        p = b + k - 1
        while p < e:
            ends.append(p)
            p += segment_size  # the k-mer restarts after a splitter: next candidate segment_size symbols later
        if ends[-1] != e - 1:
            ends.append(e - 1)  # right-most k-mer of the tail
    ends_t = torch.tensor(ends, dtype=torch.int64, device=ref.device)
    can, _d, _rc = kmers_ending_at(ref, ends_t, k)
    return np.unique(can)


_LETTERS = None


def make_fasta(codes, ctg_off, names, width=60):
    """codes (uint8 tensor in HBM, contigs back to back at ctg_off) -> the bytes of the FASTA FILE that holds them, resident in
    HBM: a header line per contig, `width` letters per line, '\\n' line ends.  Returns (raw uint8 tensor with 64 bytes of slack,
    n_raw, raw_begin, raw_end): contig c's sequence lines are raw[raw_begin[c]:raw_end[c]] -- what genome_io hands to the
    reference's workers (src/core/agc_compressor.cpp:2160-2228), what agc_hip_pack_fasta_* takes."""
    global _LETTERS
    dev = codes.device
    if _LETTERS is None or _LETTERS.device != dev:
        _LETTERS = torch.tensor(list(b"ACGTNRYSWKMBDHVU"), dtype=torch.uint8, device=dev)
    n_ctg = len(ctg_off) - 1
    heads = [(">%s\n" % names[c]).encode() for c in range(n_ctg)]
    lens = [int(ctg_off[c + 1]) - int(ctg_off[c]) for c in range(n_ctg)]
    body = [ln + (ln + width - 1) // width for ln in lens]
    n_raw = sum(len(h) for h in heads) + sum(body)
    raw = torch.empty(n_raw + 64, dtype=torch.uint8, device=dev)
    raw[n_raw:] = 0
    rb, re_ = np.zeros(n_ctg, np.uint64), np.zeros(n_ctg, np.uint64)
    o = 0
    step = (1 << 26) // width * width
    for c in range(n_ctg):
        raw[o:o + len(heads[c])] = torch.tensor(list(heads[c]), dtype=torch.uint8, device=dev)
        o += len(heads[c])
        rb[c] = o
        b0 = int(ctg_off[c])
        for b in range(0, lens[c], step):
            e = min(lens[c], b + step)
            letters = _LETTERS[codes[b0 + b:b0 + e].to(torch.int64)]
            full = (e - b) // width
            if full:
                blk = raw[o:o + full * (width + 1)].view(full, width + 1)
                blk[:, :width] = letters[:full * width].view(full, width)
                blk[:, width] = 10
                o += full * (width + 1)
            rest = (e - b) - full * width
            if rest:  # (only the contig's last line)
                raw[o:o + rest] = letters[full * width:]
                raw[o + rest] = 10
                o += rest + 1
        re_[c] = o
    assert o == n_raw
    return raw, n_raw, rb, re_
