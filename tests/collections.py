"""Seeded FASTA collections for whole-archive parity (reference CLI vs agc_amd)."""
import os

import numpy as np

from agc_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
TOY = os.path.join(HERE, "golden", "toy_ex")

# name -> (cli args, builder)
CONFIGS = {
    "toy_c1": (["-k", "25", "-l", "17"], "toy"),
    "toy_default": ([], "toy"),
    # 4 contigs x 250 kb, 6 samples, SNPs only: both-splitter path + missing-middle splits
    "syn_snp": (["-k", "31", "-l", "20", "-s", "5000", "-b", "4"], "snp"),
    # SNPs + indels + N-runs + IUPAC, short min match, odd k
    "syn_mixed": (["-k", "25", "-l", "15", "-s", "3000", "-b", "3"], "mixed"),
    # many small genomes (BASELINE configs[1] shape, scaled): 40 x 30 kb, 1 % SNP, default segment size
    "syn_viral": (["-k", "31", "-l", "20", "-b", "7"], "viral"),
    # the same, concatenated mode
    "syn_viral_c": (["-k", "31", "-l", "20", "-b", "7", "-c"], "viral_c"),
    # contigs in a different order / missing / extra in the samples, lower-case and soft-masked input
    "syn_shuffled": (["-k", "21", "-l", "17", "-s", "2000", "-b", "50"], "shuffled"),
    # adaptive mode: samples carry contigs that have no splitter of the reference -> new splitters are
    # mined from them and used by later samples (BASELINE configs[4] shape, scaled)
    "syn_adaptive": (["-a", "-k", "31", "-l", "20", "-s", "2000", "-b", "5"], "adaptive"),
    "syn_adaptive_c": (["-a", "-c", "-k", "25", "-l", "18", "-s", "1500", "-b", "4"], "adaptive"),
    # --- twins of the BASELINE.json configs that are otherwise only benchmarked (exact CLI parameters, scaled sizes) ---
    # configs[2]: 24 contigs proportioned like GRCh38 (30 Mbp, 1/100 size), 10 samples, d = 1e-3, -k 31 -l 15 -b 100
    "syn_c3_twin": (["-k", "31", "-l", "15", "-b", "100"], "c3"),
    # configs[3] (HPP-shaped, non-adaptive, default parameters): haplotype assemblies of 300+ contigs each -- pieces of the
    # reference chromosomes in either orientation, scaffold gaps (N-runs of 1-5 kb), a few indels, unplaced short contigs
    "syn_c4_twin": ([], "c4"),
    # configs[4] (bacterial, adaptive): 64 genomes at ~5 % pairwise divergence, accessory contigs (plasmid families) that have
    # no splitter of the reference genome -> new splitters are mined and reused by later genomes
    "syn_c5_twin": (["-a", "-s", "1500"], "c5"),
}

# append plans: name -> (collection, [number of input files of each step]); step 0 is `create` (reference file
# first), every later step is one `append` of the next files; the options -a / -c of the collection carry over
APPEND_PLANS = {
    "snp_4_3": ("syn_snp", [4, 3]),
    "snp_1_3_3": ("syn_snp", [1, 3, 3]),           # reference only, then two appends
    "mixed_3_3": ("syn_mixed", [3, 3]),
    "viral_25_15": ("syn_viral", [25, 15]),        # 25 = 3 full collection batches (-b 7) + 4
    "viral_14_20_6": ("syn_viral", [14, 20, 6]),   # 14 = exactly two batches: the last batch is copied, not re-opened
    "viral_c_1_1": ("syn_viral_c", [1, 1]),
    "shuffled_2_4": ("syn_shuffled", [2, 4]),
    "adaptive_3_4": ("syn_adaptive", [3, 4]),
    "adaptive_1_2_4": ("syn_adaptive", [1, 2, 4]),
    "adaptive_c_4_3": ("syn_adaptive_c", [4, 3]),
    "toy_2_2": ("toy_c1", [2, 2]),
}


def run_append_plan(cli, plan, outdir, threads="4", env=None):
    """runs create + appends with `cli` (reference binary or agc_amd); returns the archive bytes after every step"""
    import subprocess
    coll, steps = APPEND_PLANS[plan]
    args, _ = CONFIGS[coll]
    files = build(coll, os.path.join(outdir, "in"))
    assert sum(steps) == len(files), (plan, len(files))
    carry = [a for a in args if a in ("-a", "-c")]
    out, pos, prev = [], 0, None
    for i, n in enumerate(steps):
        fn = os.path.join(outdir, f"step{i}.agc")
        if i == 0:
            cmd = [cli, "create"] + args + ["-t", threads, "-o", fn] + files[:n]
        else:
            cmd = [cli, "append"] + carry + ["-t", threads, "-o", fn, prev] + files[pos:pos + n]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert os.path.exists(fn), r.stderr[-2000:]
        out.append(open(fn, "rb").read())
        pos += n
        prev = fn
    return out


def build(name, outdir):
    """writes the FASTA files, returns their paths (reference first)"""
    kind = CONFIGS[name][1]
    os.makedirs(outdir, exist_ok=True)
    if kind == "toy":
        return [os.path.join(TOY, f) for f in ("ref.fa", "a.fa", "b.fa", "c.fa")]
    rng = np.random.default_rng({"snp": 11, "mixed": 12, "viral": 13, "viral_c": 13, "shuffled": 14, "adaptive": 15,
                                 "c3": 16, "c4": 17, "c5": 18}[kind])
    files = []

    def write(fn, contigs, names):
        p = os.path.join(outdir, fn)
        synth.to_fasta(p, contigs, names)
        files.append(p)

    if kind == "snp":
        ref = [synth.random_seq(rng, 250_000) for _ in range(4)]
        names = [f"chr{i+1} len=250000" for i in range(4)]
        write("ref.fa", ref, names)
        for s in range(6):
            write(f"s{s}.fa", [synth.mutate(rng, c, 0.002) for c in ref], names)
    elif kind == "mixed":
        ref = [synth.random_seq(rng, int(n)) for n in (180_000, 90_000, 20_000, 40)]
        names = ["ctgA", "ctgB extra words here", "ctgC", "tiny"]
        write("ref.fa", ref, names)
        for s in range(5):
            write(f"m{s}.fa", [synth.mutate(rng, c, 0.004, n_runs=2, iupac=3, indels=2) if c.size > 100 else c.copy() for c in ref], names)
    elif kind in ("viral", "viral_c"):
        ref = synth.random_seq(rng, 30_000)
        genomes = [ref] + [synth.mutate(rng, ref, 0.01) for _ in range(39)]
        if kind == "viral":
            for i, g in enumerate(genomes):
                write(f"g{i:03d}.fa", [g], [f"MN{i:05d}.1 genome {i}"])
        else:
            write("ref.fa", [genomes[0]], ["MN00000.1 genome 0"])
            write("all.fa", genomes[1:], [f"MN{i:05d}.1 genome {i}" for i in range(1, 40)])
    elif kind == "shuffled":
        ref = [synth.random_seq(rng, int(n)) for n in (60_000, 50_000, 45_000, 30_000, 10_000)]
        names = [f"c{i}" for i in range(5)]
        write("ref.fa", ref, names)
        for s in range(4):
            order = rng.permutation(5)
            keep = order[: 3 + s % 3]
            ctg = [synth.mutate(rng, ref[i], 0.003, n_runs=1) for i in keep]
            nm = [names[i] for i in keep]
            ctg.append(synth.random_seq(rng, 25_000))  # a contig unrelated to the reference
            nm.append(f"novel{s}")
            write(f"x{s}.fa", ctg, nm)
        # a soft-masked copy of a sample: lower-case bases compress identically
        p = os.path.join(outdir, "x0.fa")
        q = os.path.join(outdir, "x0lower.fa")
        with open(p, "rb") as f:
            lines = f.read().split(b"\n")
        with open(q, "wb") as f:
            f.write(b"\n".join(l if l.startswith(b">") else l.lower() for l in lines))
        files.append(q)
    elif kind == "adaptive":
        ref = [synth.random_seq(rng, int(n)) for n in (50_000, 30_000)]
        write("ref.fa", ref, ["r0", "r1"])
        novel = [synth.random_seq(rng, int(n)) for n in (40_000, 25_000, 1_200, 60_000)]
        plan = [([0, 1], [0]), ([0], [0, 1]), ([1], [1, 2]), ([0, 1], [0, 3]), ([], [3, 1]), ([1], [2, 0, 3])]
        for s_, (rc_, nv_) in enumerate(plan):
            ctg = [synth.mutate(rng, ref[i], 0.004) for i in rc_] + [synth.mutate(rng, novel[i], 0.004, n_runs=1) for i in nv_]
            nm = [f"s{s_}_r{i}" for i in rc_] + [f"s{s_}_n{i}" for i in nv_]
            write(f"a{s_}.fa", ctg, nm)
    elif kind == "c3":
        from agc_amd import synth_dev
        ln = synth_dev.contig_lengths(30_000_000)
        ref = [synth.random_seq(rng, int(n)) for n in ln]
        names = [f"chr{i + 1}" for i in range(len(ln))]
        write("GRCh38_twin.fa", ref, names)
        for s_ in range(10):
            write(f"asm{s_:02d}.fa", [synth.mutate(rng, c, 1e-3) for c in ref], names)
    elif kind == "c4":
        from agc_amd import synth_dev
        ln = synth_dev.contig_lengths(20_000_000)
        ref = [synth.random_seq(rng, int(n)) for n in ln]
        write("CHM13_twin.fa", ref, [f"chr{i + 1}" for i in range(len(ln))])
        for s_ in range(6):
            ctg, nm = [], []
            for ci, c in enumerate(ref):
                hap = synth.mutate(rng, c, 1e-3, indels=3)
                # assembly contigs: the haplotype cut at random breakpoints (about 13 pieces per chromosome)
                cuts = np.sort(rng.integers(0, hap.size, size=int(rng.integers(10, 16))))
                b = 0
                for e in list(cuts) + [hap.size]:
                    piece = hap[b:int(e)].copy()
                    b = int(e)
                    if piece.size == 0:
                        continue
                    if rng.random() < 0.5:  # assemblers report either strand
                        piece = piece[::-1].copy()
                        m = piece < 4
                        piece[m] = 3 - piece[m]
                    if piece.size > 200_000 and rng.random() < 0.3:  # scaffold gap
                        p = int(rng.integers(10_000, piece.size - 10_000))
                        piece[p:p + int(rng.integers(1_000, 5_000))] = 4
                    ctg.append(piece)
                    nm.append(f"HG{s_:05d}#1#JAHBC{len(ctg):07d}.1")
            for _ in range(20):  # unplaced short contigs, unrelated to the reference
                ctg.append(synth.random_seq(rng, int(rng.integers(500, 20_000))))
                nm.append(f"HG{s_:05d}#1#JAHBD{len(ctg):07d}.1")
            order = rng.permutation(len(ctg))
            write(f"HG{s_:05d}.1.fa", [ctg[i] for i in order], [nm[i] for i in order])
    elif kind == "c5":
        anc = [synth.random_seq(rng, 120_000)]
        plasmids = [synth.random_seq(rng, int(rng.integers(2_000, 9_000))) for _ in range(12)]
        write("K12_twin.fa", [synth.mutate(rng, anc[0], 0.025)], ["NC_000913.3 chromosome"])
        for s_ in range(63):
            ctg = [synth.mutate(rng, anc[0], 0.025, indels=2)]
            nm = [f"NZ_CP{s_:06d}.1 strain {s_} chromosome"]
            for pi in rng.permutation(12)[: int(rng.integers(0, 3))]:
                ctg.append(synth.mutate(rng, plasmids[int(pi)], 0.025))
                nm.append(f"NZ_CP{s_:06d}p{int(pi)}.1 strain {s_} plasmid p{int(pi)}")
            write(f"GCF_{s_:09d}.fa", ctg, nm)
    return files
