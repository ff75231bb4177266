"""Seeded (reference, text) cases shared by the oracle and GPU parity tests.
Mirrors the coverage of SURVEY.md App. B.8: identical text, SNPs, SNPs + N-runs + IUPAC +
indels, unrelated text, N-runs inside the reference, low-complexity reference that
saturates the 64-probe limit, short and u32-table (ref/4 >= 65535) regimes."""
import numpy as np

from agc_amd import synth


def lz_cases(seed=7, n_cases=60, lengths=(50, 200, 1000, 5000, 20000), mmls=(15, 17, 20, 24, 32)):
    rng = np.random.default_rng(seed)
    out = []
    for case in range(n_cases):
        mml = int(rng.choice(mmls))
        L = int(rng.choice(lengths))
        kind = case % 8
        ref = synth.random_seq(rng, L)
        if kind == 4:
            unit = synth.random_seq(rng, int(rng.integers(3, 9)))
            ref = np.tile(unit, L // unit.size + 1)[:L].copy()
        if kind == 5:
            ref = synth.mutate(rng, ref, 0, n_runs=3, iupac=3)
        if kind == 0:
            text = ref.copy()
        elif kind == 1:
            text = synth.mutate(rng, ref, 0.01)
        elif kind == 2:
            text = synth.mutate(rng, ref, 0.05, n_runs=3, iupac=4, indels=3)
        elif kind == 3:
            text = synth.random_seq(rng, L)
        elif kind == 6:
            text = synth.mutate(rng, ref, 0.001, indels=1)
            text[:40] = 4          # leading N-run
            text[-30:] = 4         # trailing N-run
        elif kind == 7:
            # text = suffix of the reference (match runs to the end of both)
            text = synth.mutate(rng, ref[L // 3:], 0.002)
        else:
            text = synth.mutate(rng, ref, 0.02, n_runs=2, indels=2)
        out.append((mml, ref, text))
    return out
