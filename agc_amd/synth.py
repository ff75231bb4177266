"""Synthetic genome collections for tests and bench (SURVEY.md §8d generator spec):
uniform i.i.d. ACGT reference; each sample = reference with independent per-base
substitution at rate d (substituted base uniform among the other three).
Sequences are symbol codes (A0 C1 G2 T3 N4 ...), the layout after
preprocess_raw_contig (src/core/agc_compressor.cpp:907-951)."""
import numpy as np


def random_seq(rng, n):
    return rng.integers(0, 4, size=n, dtype=np.uint8)


def mutate(rng, seq, d, n_runs=0, iupac=0, indels=0):
    """substitutions at rate d; optional N-runs, IUPAC codes and indels."""
    s = seq.copy()
    if d > 0:
        m = rng.random(s.size) < d
        s[m] = (s[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
    for _ in range(n_runs):
        if s.size < 8:
            break
        p = int(rng.integers(0, s.size - 4))
        ln = int(rng.integers(1, 40))
        s[p:p + ln] = 4
    for _ in range(iupac):
        if s.size:
            s[int(rng.integers(0, s.size))] = rng.integers(5, 16)
    for _ in range(indels):
        if not s.size:
            break
        p = int(rng.integers(0, s.size))
        if rng.random() < 0.5:
            s = np.concatenate([s[:p], random_seq(rng, int(rng.integers(1, 30))), s[p:]])
        else:
            s = np.concatenate([s[:p], s[p + int(rng.integers(1, 30)):]])
    return s


CODE2ASCII = np.frombuffer(b"ACGTNRYSWKMBDHVU", dtype=np.uint8)


def to_fasta(path, contigs, names, width=80):
    """contigs: list of code arrays (codes < 16); 80-column lines, vectorised."""
    with open(path, "wb") as f:
        for name, c in zip(names, contigs):
            f.write(b">" + name.encode() + b"\n")
            a = CODE2ASCII[c]
            full = a.size // width * width
            if full:
                body = np.empty((full // width, width + 1), np.uint8)
                body[:, :width] = a[:full].reshape(-1, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if a.size > full:
                f.write(a[full:].tobytes() + b"\n")
