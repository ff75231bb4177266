#!/usr/bin/env python3
"""bench.py -- AGC `create` hot path on MI355X: input Gbp/s compressed.

One "step" = one pass of the hot path (FASTA bytes -> 2-bit words -> splitter scan -> add_segment
classification -> LZ-diff encode of every placed segment against its group reference -> pack
bookkeeping) over one synthetic human-scale sample that is resident in HBM as the bytes of its
FASTA file (BASELINE.json configs[2]: GRCh38-shaped reference, 0.1 % divergence, k=31 l=15 b=100;
--prepacked: already in the 2-bit layout when the timer starts, round 4's region);
the zstd packing the steps defer (packs flush every b samples) runs in Close(), which is
inside the timed region.  The step is the code path `agc_amd create` runs -- AddSampleFiles converts a file's
bytes with the same kernels (agc_hip_sample_pack_fasta) -- and whose archives are byte-identical to the reference's
(tests/test_gpu_archive.py; this very input at full size: test_configs2_at_full_size_equals_the_reference_archive).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` = bases of all ranks / max-over-ranks time.
roofline.* is for the kernel with the largest time per step (every kernel of the path competes), timed with HIP
events on the library's own stream, on the layout as built and on SURVEY 8d's 2-bit figures; cpu_baseline is the reference CPU implementation (oracle/_ref/agc, when it was
prebuilt) or the oracle port, timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

# (before anything initialises HIP: the runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues, default 4; the
# library has eight -- agc_amd/csrc/api.hip: agc_hip_create -- and on four they wait for one another)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, MML, SEG, PACK = 31, 15, 60000, 100
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_SYMBOL = 0.25  # the layout every kernel of the step reads: samples AND group references in 2-bit words (sym_view.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gbp", type=float, default=3.0, help="bases per sample (Gbp)")
    ap.add_argument("--div", type=float, default=1e-3, help="per-base substitution rate")
    ap.add_argument("--cpu-baseline-mbp", type=float, default=300.0, help="size of the CPU baseline sample (Mbp per genome)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--positional-splitters", action="store_true",
                    help="skip determine_splitters: take the k-mer at every segment_size-th position (valid for an i.i.d. reference)")
    ap.add_argument("--pack-cardinality", type=int, default=0, metavar="B",
                    help="segments per delta pack (agc -b; default: BASELINE configs[2]'s 100).  With B below the number of samples "
                         "packs fill DURING the steps and the entropy stage runs beside them: the overlap variant (VERDICT r1 item 6)")
    ap.add_argument("--threads", type=int, default=0, help="host threads for libzstd (default: all cores / n_gpus)")
    ap.add_argument("--from-fasta", type=int, default=0, metavar="K",
                    help="FILE MODE instead of the HBM-resident bench: write the reference + K samples as FASTA (tmpfs), run "
                         "create through AddSampleFiles (read + a1 on the GPU + the whole path + archive to tmpfs) and report that rate")
    ap.add_argument("--shards", action="store_true",
                    help="N > 1: one independent archive shard per rank (no collective) instead of the default: all ranks feed ONE "
                         "archive (ordered commit from broadcast commit records, entropy stage spread over the ranks' GPUs; agc_amd/dist.py)")
    ap.add_argument("--single-archive", action="store_true", help="(the default for N > 1; kept for older command lines)")
    ap.add_argument("--c5-samples", type=int, default=256, help="--config c5slice: number of 5 Mbp genomes")
    ap.add_argument("--config", default="c2", choices=["c2", "c1", "c4twin", "c5twin", "c5slice"],
                    help="c2 (default): BASELINE configs[2], the headline (HBM-resident 3 Gbp samples).  c1: BASELINE configs[1] -- 1000 "
                         "SARS-CoV-2-size genomes (30 kb, 1 %% SNP from one reference), default parameters, from FASTA files through the product "
                         "CLI path; the reference CLI is timed beside it on the same files and the two archives are compared.  c4twin / c5twin: the "
                         "1/100 twins of configs[3] / configs[4] the parity tests use, the same way; c5slice: a slice of configs[4] at FULL contig size -- "
                         "--c5-samples (256) bacterial-size genomes of 5 Mbp, 5 %% pairwise divergence, plasmid families without a splitter of the "
                         "reference, adaptive mode (-a), default parameters.  All with a stage breakdown and per-kernel roofline rows")
    ap.add_argument("--prepacked", action="store_true",
                    help="round-4 input: every sample packed into the 2-bit layout BEFORE the timer starts.  Default since round 5: the "
                         "samples are resident in HBM as the bytes of their FASTA files and the timed step turns them into the 2-bit layout "
                         "(agc_hip_pack_fasta_*: preprocess_raw_contig + packing in one pass), two samples ahead on a stream of its own")
    ap.add_argument("--fasta-width", type=int, default=60, help="letters per FASTA line of the HBM-resident inputs")
    ap.add_argument("--no-prefetch", action="store_true", help="do not announce the next sample (its scan then runs inside its own step)")
    ap.add_argument("--verify-entropy", action="store_true",
                    help="CHECKING RUN, not a measurement: every frame the device entropy stage returns (all the packs of this run's Close) is "
                         "compressed again by the host's libzstd 1.4.9 at level 17 and compared byte for byte (AGC_AMD_VERIFY_DEV_FRAMES); "
                         "Close() fails on a difference.  The line's `value` includes that work -- keep the log, not the number")
    return ap.parse_args()


def cgroup_cpu_stat():
    """usage / throttling counters of this container's CPU quota (cgroup v2 cpu.stat), {} when unreadable"""
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except Exception:
        return {}


def host_cpus():
    """CPUs this container may actually use: the cgroup quota when there is one (the GPU box reports
    256 logical CPUs but grants 16), else the logical CPU count."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


PMC_SUMMARY = next((p_ for p_ in (os.path.join("profiles", r_, "pmc_summary.csv") for r_ in ("r6", "r5", "r4", "r3"))
                    if os.path.exists(os.path.join(ROOT, p_))), os.path.join("profiles", "r6", "pmc_summary.csv"))
KERNEL_SYMBOL = {"scan": "agc::scan_packed_kernel", "encode": "agc::lz_parse_kernel<0>", "estimate": "agc::lz_parse_kernel<1>",
                 "costvec": "agc::lz_parse_kernel<2>", "filter": "agc::key_filter_kernel", "pack": "agc::pack_fasta_kernel",
                 "zstd": "agc::zstd_frames_grp_kernel<3, 2>"}
# a row whose time covers more than one kernel: their counters are added up (the conversion = counting pass + pack pass; the scans
# of the tile counts between them move a few hundred KB)
KERNEL_SYMBOLS_ALL = {"pack": ("agc::pack_fasta_count_kernel", "agc::pack_fasta_kernel")}
# bytes per symbol each kernel reads in the layout AS BUILT = SURVEY 8d's 2-bit column since round 4: scan, key filter and the
# three LZ parses read texts and references as 2-bit words where they lie (no expansion, no reverse-complement staging)
# (pack: the FASTA bytes of a symbol, 1 + 1 / line width, read TWICE -- the counting pass and the pack pass, pack_kernels.hip -- + its
# 0.25 B written: the layout as built; SURVEY 8d's figure reads them once.  Filled in by main() for --fasta-width)
AS_BUILT_BPS = {"scan": 0.25, "encode": 0.25, "estimate": 0.25, "costvec": 0.25, "filter": 0.25, "pack": 2.0 * (1.0 + 1.0 / 60) + 0.25}
PACKED_BPS = dict(AS_BUILT_BPS, pack=1.0 + 1.0 / 60 + 0.25)


def pmc_table(path=None):
    """{kernel symbol: {counter: max KB per dispatch}} from the committed rocprofv3 PMC summary of this workload
    (scripts/pmc_summary.py over separate --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 2 --warmup 1`; max over
    dispatches = the full-size launches).  Empty when the summary is absent."""
    tab = {}
    try:
        for line in open(os.path.join(ROOT, path or PMC_SUMMARY)):
            f = line.strip().rsplit(",", 4)  # (a kernel name may hold commas: template arguments)
            if len(f) >= 5 and f[0] != "kernel":
                tab.setdefault(f[0], {})[f[1]] = float(f[4])
                tab.setdefault(f[0], {})[f[1] + ":sum"] = float(f[3]) * float(f[2])  # (mean x dispatches)
    except OSError:
        pass
    return tab


# Kernels whose loads are narrow and scattered (one 4-byte word per lane in a line of its own: the zstd match finder's hash /
# chain / tree tables).  profiles/r4/fetch_calibration.txt: calib_probe4 (134 M such loads in a 4 GiB table) counts 63.9 B per
# load, calib_probe4_pair (the same plus the word 64 B further in the same 128-B line) 1.71 x that -- the memory side serves them
# as 64-B requests and FETCH_SIZE counts those as they are; the x2 of the guide belongs to wide coalesced loads (calib_stream:
# 0.50 of the bytes asked for).
NARROW_LOADS = {"zstd"}


def pmc_traffic(tab, name, per_run=False):
    """HBM bytes per launch: FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 tallies the 128-B requests of wide coalesced
    loads as 64 B) + WRITE_SIZE -- except for the kernels of NARROW_LOADS, whose requests the counter tallies as they are
    (this repo's calibration, see above): FETCH_SIZE + WRITE_SIZE.  pmc_traffic_range() states both sums for every kernel: a
    kernel that mixes streaming reads with table probes (the LZ parses) lies in between."""
    tot = 0
    for sym in KERNEL_SYMBOLS_ALL.get(name, (KERNEL_SYMBOL[name],)):
        c = tab.get(sym)
        if not c or "FETCH_SIZE" not in c:
            return None
        if per_run:  # (the --config rows: all dispatches of a run together, like their time)
            tot += int(((1.0 if name in NARROW_LOADS else 2.0) * c["FETCH_SIZE:sum"] + c.get("WRITE_SIZE:sum", 0.0)) * 1024)
        else:
            tot += int(((1.0 if name in NARROW_LOADS else 2.0) * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024)
    return tot


def pmc_traffic_range(tab, name):
    """[FETCH_SIZE + WRITE_SIZE, 2 x FETCH_SIZE + WRITE_SIZE] in bytes per launch (see pmc_traffic)"""
    lo = hi = 0
    for sym in KERNEL_SYMBOLS_ALL.get(name, (KERNEL_SYMBOL[name],)):
        c = tab.get(sym)
        if not c or "FETCH_SIZE" not in c:
            return None
        lo += int((1.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024)
        hi += int((2.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024)
    return [lo, hi]


def cpu_baseline(args, mbp):
    """reference CPU implementation on a bounded twin of the workload (same generator, smaller):
    the per-sample path of the reference CLI: samples appended to an archive that holds the reference (see below), all host cores."""
    from agc_amd import synth
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "agc")
    rng = np.random.default_rng(12345)
    n = int(mbp * 1e6)
    ctg_len = [n // 4] * 4
    refc = [synth.random_seq(rng, l) for l in ctg_len]
    n_samples = 8
    cores = host_cpus()
    if os.path.exists(ref_bin):
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            names = [f"chr{i+1}" for i in range(4)]
            synth.to_fasta(os.path.join(td, "ref.fa"), refc, names)
            files = []
            for s in range(n_samples):
                smp = [synth.mutate(rng, c, args.div) for c in refc]
                fn = os.path.join(td, f"s{s}.fa")
                synth.to_fasta(fn, smp, names)
                files.append(fn)
            t_threads = str(max(1, min(cores, 128)))
            common = [ref_bin, "create", "-k", str(K), "-l", str(MML), "-b", str(PACK), "-s", str(SEG), "-t", t_threads, "-o"]

            # The per-sample path in isolation: the archive with the reference is made once (untimed: `create` spends 20+ s in the
            # reference's own preprocessing at this size, and the noise of two such walls swamps the difference of a few seconds
            # the samples make); the samples are then APPENDED -- the same compress_contig / add_segment / store path
            # (agc_compressor.cpp:2330-2374) -- and wall(append all) - wall(append one) is the time of n_samples - 1 samples.
            # Three repetitions on the same files, medians.
            base = os.path.join(td, "base.agc")
            subprocess.run(common + [base, os.path.join(td, "ref.fa")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

            def run(fl, threads):
                t0 = time.time()
                subprocess.run([ref_bin, "append", "-t", threads, "-o", os.path.join(td, "o.agc"), base] + fl,
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                return time.time() - t0
            med = lambda xs: sorted(xs)[len(xs) // 2]
            ones_ = [run(files[:1], t_threads) for _ in range(3)]
            alls_ = [run(files, t_threads) for _ in range(3)]
            t_one, t_all = med(ones_), med(alls_)
            dt = max(t_all - t_one, 1e-6)
            spread = (max(alls_) - min(alls_)) / max(dt, 1e-6)
            # SURVEY 8d asks for T in {1, all cores}: the single-thread figure on two samples against one
            t1_one = run(files[:1], "1")
            t1_two = run(files[:2], "1")
            dt1 = max(t1_two - t1_one, 1e-6)
            model = "?"
            try:
                model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
            except Exception:
                pass
            return {"value": (n_samples - 1) * n / dt / 1e9, "unit": "Gbp/s", "cores": int(t_threads), "kind": "reference",
                    "sample": f"oracle/_ref/agc append -t {t_threads} to an archive holding the {mbp:g} Mbp reference: median of 3 of wall({n_samples} samples, "
                              f"d={args.div:g}) - median of 3 of wall(1 sample) = {t_all:.2f} - {t_one:.2f} = {dt:.2f} s for {n_samples - 1} samples",
                    "cpu_model": model, "walls_all_s": [round(x, 2) for x in alls_], "walls_one_s": [round(x, 2) for x in ones_],
                    "spread": round(spread, 3),
                    "value_1_thread": round(n / dt1 / 1e9, 4),
                    "sample_1_thread": f"the same with -t 1: wall(2 samples) - wall(1 sample) = {dt1:.2f} s"}
    # oracle port, single thread: scan + encode of one sample
    from oracle import agc_oracle as O
    spl = O.determine_splitters(refc, K, SEG)
    smp = [synth.mutate(rng, c, args.div) for c in refc]
    t0 = time.time()
    for rc_, sc in zip(refc, smp):
        s = O.scan_contig(sc, K, spl)
        r = O.scan_contig(rc_, K, spl)
        for a, l in zip(s["start"], s["len"]):
            z = O.LZ(rc_[int(a):int(a) + int(l)], MML)
            z.encode(sc[int(a):int(a) + int(l)])
    dt = time.time() - t0
    return {"value": n / dt / 1e9, "unit": "Gbp/s", "cores": 1, "kind": "port",
            "sample": f"oracle/agc_oracle.c scan + index + encode of one {mbp:g} Mbp sample, {dt:.2f} s"}


def config_cli(args):
    """The BASELINE configs that are not the headline, as whole `agc_amd create` CLI runs beside the reference CLI on the same files:
      c1     -- configs[1]: 1000 genomes x 30 kb, 1 % SNP from one reference, AGC's defaults, one FASTA file per genome;
      c4twin -- the 1/100 twin of configs[3] the parity tests use (tests/collections.py: syn_c4_twin, HPP-shaped, default parameters);
      c5twin -- the twin of configs[4] (syn_c5_twin: 64 bacterial genomes, 5 % divergence, adaptive mode -a).
    The line carries a stage breakdown -- the compressor's own -v 1 stage seconds (file reading, scan, classification, registration,
    encode, bookkeeping, entropy stage split into device / host pool), and the fixed cost of a run (process start, HIP context, libzstd:
    measured with a one-contig archive) -- so that the terms of these host-bound shapes are on the table; `value` is the whole run,
    `value_without_start` excludes the fixed cost for both programs."""
    import hashlib
    import re
    from agc_amd import synth
    threads = args.threads or host_cpus()
    amd = os.path.join(ROOT, "agc_amd", "bin", "agc_amd")
    refbin = os.path.join(ROOT, "oracle", "_ref", "agc")
    ref_env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        if args.config == "c1":
            n = 1000
            rng = np.random.default_rng(2)
            ref = synth.random_seq(rng, 30_000)
            files, cli_args = [], []
            for i in range(n):
                g = ref if i == 0 else synth.mutate(rng, ref, 0.01)
                fn = os.path.join(td, f"g{i:04d}.fa")
                synth.to_fasta(fn, [g], [f"MN{i:06d}.1 synthetic genome {i}"])
                files.append(fn)
            what = ("BASELINE configs[1]: 1000 synthetic SARS-CoV-2-size genomes (30 kb, 1 % SNP from one reference), k=31 l=20 s=60000 b=50, "
                    "one FASTA file each (tmpfs)")
        elif args.config == "c5slice":
            # configs[4] at full contig size, a slice of its 50 k genomes: every genome = the ancestor's 5 Mbp chromosome at 2.5 %
            # (5 % between two genomes) + 0-2 plasmids of 12 families the reference genome has no splitter for (adaptive mode mines them)
            n = max(2, args.c5_samples)
            rng = np.random.default_rng(5)
            anc = synth.random_seq(rng, 5_000_000)
            plasmids = [synth.random_seq(rng, int(rng.integers(20_000, 90_000))) for _ in range(12)]
            files, cli_args = [], ["-a"]
            for i in range(n):
                ctg, nm = [synth.mutate(rng, anc, 0.025)], [f"NZ_CP{i:06d}.1 strain {i} chromosome"]
                if i:
                    for pi in rng.permutation(12)[: int(rng.integers(0, 3))]:
                        ctg.append(synth.mutate(rng, plasmids[int(pi)], 0.025))
                        nm.append(f"NZ_CP{i:06d}p{int(pi)}.1 strain {i} plasmid p{int(pi)}")
                fn = os.path.join(td, f"GCF_{i:09d}.fa")
                synth.to_fasta(fn, ctg, nm)
                files.append(fn)
            what = (f"slice of BASELINE configs[4] at full contig size: {n} synthetic bacterial genomes (5 Mbp chromosome at 2.5 % from one ancestor = 5 % "
                    "pairwise, 0-2 plasmids of 12 families), adaptive mode `-a`, k=31 l=20 s=60000 b=50, one FASTA file each (tmpfs)")
        else:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from tests import collections as C
            name = {"c4twin": "syn_c4_twin", "c5twin": "syn_c5_twin"}[args.config]
            cli_args, _ = C.CONFIGS[name]
            files = C.build(name, os.path.join(td, "in"))
            what = (f"twin of BASELINE configs[{3 if args.config == 'c4twin' else 4}] at 1/100 size (tests/collections.py: {name}, "
                    f"`{' '.join(cli_args) or 'default parameters'}`), {len(files)} FASTA files (tmpfs)")
        bases = 0
        for fn in files:  # (letters = bytes - line ends - header lines)
            raw = np.fromfile(fn, np.uint8)
            nl = np.flatnonzero(raw == 10)
            hdr = sum(int(nl[np.searchsorted(nl, st_)]) - int(st_) for st_ in np.flatnonzero(raw == ord(">")))
            bases += int(raw.size - nl.size - (raw == 13).sum()) - hdr
        tiny = os.path.join(td, "tiny.fa")
        synth.to_fasta(tiny, [synth.random_seq(np.random.default_rng(1), 2000)], ["tiny"])

        def run(binary, out, fl, extra=(), env=None):
            t0 = time.perf_counter()
            r = subprocess.run([binary, "create"] + list(cli_args) + list(extra) + ["-t", str(threads), "-o", out] + fl, check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, env=env)
            return time.perf_counter() - t0, r.stderr.decode(errors="replace")
        run(amd, os.path.join(td, "w.agc"), files)  # warm-up (page cache, HIP code objects)
        reps = args.steps if args.steps > 1 else 3
        walls = []
        for _ in range(reps):
            walls.append(run(amd, os.path.join(td, "a.agc"), files, ("-v", "1")))
        walls.sort(key=lambda x: x[0])
        t_amd, err = walls[len(walls) // 2]
        t_fixed = sorted(run(amd, os.path.join(td, "t.agc"), [tiny])[0] for _ in range(3))[1]
        stages = {}
        m = re.search(r"seconds: (.*?) \(inside the device library: ([0-9.e+-]+)\)", err)
        if m:
            toks = m.group(1).split()
            stages = {toks[i]: float(toks[i + 1]) for i in range(0, len(toks) - 1, 2)}
            stages["inside_device_library"] = float(m.group(2))
        m = re.search(r"windows (\d+) commit-runs (\d+) revalidated (\d+) windows-cut (\d+)", err)
        if m:
            stages["windows"], stages["commit_runs"], stages["revalidated_segments"], stages["windows_cut"] = (int(x) for x in m.groups())
        m = re.search(r"entropy-seconds: host-pool ([0-9.e+-]+) device ([0-9.e+-]+) staging ([0-9.e+-]+) caller-waited ([0-9.e+-]+)", err)
        if m:
            stages["entropy_host_pool_s"], stages["entropy_device_call_s"], stages["entropy_staging_s"], stages["entropy_caller_waited_s"] = (float(x) for x in m.groups())
        m = re.findall(r"entropy stage: device ([0-9.e+-]+) MB in ([0-9.e+-]+) s, host ([0-9.e+-]+) MB in ([0-9.e+-]+) s", err)
        if m:
            stages["entropy_device_mb"] = round(sum(float(x[0]) for x in m), 2)
            stages["entropy_device_s"] = round(sum(float(x[1]) for x in m), 3)
            stages["entropy_host_mb"] = round(sum(float(x[2]) for x in m), 2)
            stages["entropy_host_s"] = round(sum(float(x[3]) for x in m), 3)
        # per-kernel rows: one more run with HIP events around every kernel (AGC_AMD_KERNEL_TIMES=1: every launch is waited for,
        # so this run's wall time is not a measurement; the rows' times are the kernels' own)
        kern, dominant = {}, None
        _t, kerr = run(amd, os.path.join(td, "k.agc"), files, ("-v", "1"), env=dict(os.environ, AGC_AMD_KERNEL_TIMES="1"))
        mk = re.search(r"kernel-ms:(.*)", kerr)
        ms_ = re.search(r"kernel-symbols:(.*)", kerr)
        if mk and ms_:
            t_ = mk.group(1).split()
            kms = {t_[i]: (float(t_[i + 1]), int(t_[i + 2])) for i in range(0, len(t_) - 2, 3)}
            y_ = ms_.group(1).split()
            sym = {y_[i]: int(y_[i + 1]) for i in range(0, len(y_) - 1, 2)}
            tab = pmc_table(next((p_ for p_ in (os.path.join("profiles", r_, f"pmc_summary_{args.config}.csv") for r_ in ("r6", "r5"))
                                  if os.path.exists(os.path.join(ROOT, p_))), os.path.join("profiles", "r6", f"pmc_summary_{args.config}.csv")))
            for name in ("scan", "encode", "estimate", "costvec", "filter", "zstd"):
                if name not in kms or not kms[name][0]:
                    continue
                ms, n_l = kms[name]
                alg = (sym["zstd_dev_in"] + sym["zstd_dev_out"]) if name == "zstd" else 0.25 * sym.get(name, 0)
                ach = alg / (ms * 1e-3) / 1e9
                tr = pmc_traffic(tab, name, per_run=True)
                kern[name] = {"ms_per_run": round(ms, 3), "launches": n_l, "kernel": KERNEL_SYMBOL.get(name),
                              "algorithmic_bytes": int(alg), "achieved": round(ach, 3), "frac": round(ach / HBM_PEAK_GBS, 6),
                              "traffic": tr, "waste": round(tr / alg, 2) if tr and alg else None}
            other = {k_: round(v[0], 3) for k_, v in kms.items() if k_ not in kern and v[0]}
            if kern:
                dominant = max(kern, key=lambda k_: kern[k_]["ms_per_run"])
            kern["_other_ms_per_run"] = other  # (index build, segments, reference store, preprocess: latency-bound helpers)
        cpu = None
        if os.path.exists(refbin) and not args.no_cpu_baseline:
            ref_reps = 1 if args.config == "c5slice" else 3  # (c5slice: minutes of reference-CLI time per run)
            rt = sorted(run(refbin, os.path.join(td, "r.agc"), files, env=ref_env)[0] for _ in range(ref_reps))[ref_reps // 2]
            rt_fixed = sorted(run(refbin, os.path.join(td, "rt.agc"), [tiny], env=ref_env)[0] for _ in range(3))[1]
            same = hashlib.sha256(open(os.path.join(td, "a.agc"), "rb").read()).digest() == hashlib.sha256(open(os.path.join(td, "r.agc"), "rb").read()).digest()
            cpu = {"value": round(bases / rt / 1e9, 4), "unit": "Gbp/s", "cores": threads, "kind": "reference",
                   "sample": f"oracle/_ref/agc create -t {threads} on the same {len(files)} files: median of {ref_reps} = {rt:.3f} s (a one-contig archive: {rt_fixed:.3f} s)",
                   "value_without_start": round(bases / max(rt - rt_fixed, 1e-9) / 1e9, 4), "archives_identical": same}
        out = {"metric": f"input Gbp/s compressed (create), {args.config}: whole CLI run from FASTA files"
                         + (" -- A START-COST PROBE: 8 Mbp in all, the fixed start of a run is most of its wall time; read fixed_cost_s and "
                            "value_without_start, not value" if args.config == "c5twin" else ""), "value": round(bases / t_amd / 1e9, 4),
               "unit": "Gbp/s", "n_gpus": 1, "steps": reps, "warmup": 1, "ms_per_step": round(t_amd * 1e3, 1), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": what + f", `agc_amd create` (process start, HIP context, files, archive) -- median wall of {reps} runs",
                          "bases": bases, "host_threads": threads, "fixed_cost_s": round(t_fixed, 3),
                          "fixed_cost_is": "the wall time of `agc_amd create` of one 2 kb contig: process start, HIP context and streams, dlopen of libzstd, archive",
                          "value_without_start": round(bases / max(t_amd - t_fixed, 1e-9) / 1e9, 4),
                          "stage_seconds": stages,
                          # the host pool's libzstd time (the entropy thread's own clock around its pool calls: they overlap the
                          # steps) as a share of the run's wall time
                          "host_entropy_share_of_wall": (round(stages["entropy_host_pool_s"] / t_amd, 3) if "entropy_host_pool_s" in stages else None)},
               "roofline": ({"bound": "hbm", "kernel": kern[dominant]["kernel"], "achieved": kern[dominant]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": kern[dominant]["frac"], "traffic": kern[dominant]["traffic"],
                             "dominant_by": "largest kernel time per run (HIP events around every launch, a separate run with AGC_AMD_KERNEL_TIMES=1)",
                             "algorithmic_bytes": "0.25 B per symbol a kernel was asked to look at (text once + reference once for the parses); zstd: packs in + frames out",
                             "kernels": kern} if dominant else None)}
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)


def file_mode(args):
    """`agc_amd create` from FASTA files at the bench's sample size: the PCIe- and file-system-inclusive rate (never `value`
    of the HBM-resident bench)."""
    import torch
    from agc_amd import host, synth, synth_dev
    dev = torch.device("cuda:0")
    total = int(args.gbp * 1e9)
    ref, off = synth_dev.make_reference(total, 12345, dev)
    tot = int(off[-1])
    names = [f"chr{i + 1}" for i in range(len(off) - 1)]
    threads = args.threads or host_cpus()
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        def fasta(path, t):
            h = t[:tot].cpu().numpy()
            synth.to_fasta(path, [h[int(off[i]):int(off[i + 1])] for i in range(len(names))], names)
        files = [os.path.join(td, "ref.fa")]
        fasta(files[0], ref)
        for s in range(args.from_fasta):
            files.append(os.path.join(td, f"s{s}.fa"))
            fasta(files[-1], synth_dev.make_sample(ref, tot, args.div, 1000 + s, dev))
        cmp_ = host.Compressor(0)
        t0 = time.perf_counter()
        cmp_.create(os.path.join(td, "out.agc"), PACK, K, files[0], SEG, MML, n_threads=threads)
        t_create = time.perf_counter() - t0
        t0 = time.perf_counter()
        cmp_.add_sample_files([("ref", files[0])], threads)
        t_ref = time.perf_counter() - t0
        st0 = cmp_.stats()
        t0 = time.perf_counter()
        cmp_.add_sample_files([(f"s{s}", files[1 + s]) for s in range(args.from_fasta)], threads)
        t_add = time.perf_counter() - t0
        cmp_.close(threads)
        elapsed = time.perf_counter() - t0
        st1 = cmp_.stats()
        cmp_.close_handle()
        bases = st1["bases"] - st0["bases"]
        out = {"metric": "input Gbp/s compressed (create) FROM FASTA FILES (tmpfs): read + a1 on the GPU + scan + match + encode + zstd + archive",
               "value": round(bases / elapsed / 1e9, 3), "unit": "Gbp/s", "n_gpus": 1, "steps": args.from_fasta, "warmup": 0,
               "ms_per_step": round(elapsed / max(args.from_fasta, 1) * 1e3, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u8", "data": "synthetic",
               "config": {"workload": f"FILE MODE of BASELINE configs[2]: {args.from_fasta} FASTA files of {args.gbp:g} Gbp (80-column lines, tmpfs), "
                                      f"d={args.div:g}, k={K} l={MML} b={PACK}; host-resident inputs: PCIe-inclusive, NOT the HBM-resident headline",
                          "add_samples_s": round(t_add, 2), "close_s": round(elapsed - t_add, 2),
                          "setup_not_timed": f"create (reference file read + determine_splitters): {t_create:.1f} s; reference as first sample: {t_ref:.1f} s",
                          "host_threads": threads, "io_seconds": round(st1["t_io"] - st0["t_io"], 2)}}
        print(json.dumps(out), flush=True)


def _sha256_file(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    if "--verify-entropy" in sys.argv:
        os.environ["AGC_AMD_VERIFY_DEV_FRAMES"] = "1"
    args = parse()
    if args.config != "c2":
        return config_cli(args)
    if args.from_fasta:
        return file_mode(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # AGC_BENCH_ONE_GPU=1 (testing aid, a box with one GPU): every rank on cuda:0, records over gloo -- the N > 1 code path of
    # this file end to end on real kernels; the driver's runs use one GPU per rank and RCCL
    one_gpu = world > 1 and os.environ.get("AGC_BENCH_ONE_GPU") == "1"
    import datetime
    # (a collective that does not complete in ten minutes is a rank that died or a launch that does not match the node: the job
    # stops with RCCL's message instead of hanging until the driver's limit)
    if one_gpu:
        local = 0
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
    elif world > 1:
        if world > torch.cuda.device_count() or local >= torch.cuda.device_count():
            raise SystemExit(f"bench.py --gpus {world}: {torch.cuda.device_count()} GPU(s) visible to rank {rank} (LOCAL_RANK {local}); one process per GPU "
                             "(AGC_BENCH_ONE_GPU=1 runs all ranks on cuda:0 over gloo for a functional check)")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"), timeout=datetime.timedelta(seconds=600))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    red_dev = torch.device("cpu") if one_gpu else dev  # where the small reductions of the timings live
    from agc_amd import host, shard, synth_dev

    total = int(args.gbp * 1e9)
    ref, off = synth_dev.make_reference(total, 12345, dev)
    tot = int(off[-1])
    names = [f"chr{i + 1}" for i in range(len(off) - 1)]
    single = world > 1 and not args.shards
    threads = max(1, host_cpus() // world)
    if single:  # the writer rank does all the zstd work, the others need a few host threads only
        threads = max(1, host_cpus() - 2 * (world - 1)) if rank == 0 else 2
    if args.threads:
        threads = args.threads
    cmp_ = host.Compressor(local)
    if single:
        cmp_.set_distributed(rank, world, 0)
    # archive bytes are produced and discarded (out path ""): file I/O is not the path under test
    pack_card = args.pack_cardinality or PACK
    # (AGC_BENCH_ARCHIVE=<path>, a checking aid: the archive is written there and its sha256 reported in config.archive_sha256 -- two
    # runs of the same command must agree whatever AGC_AMD_ASYNC_ENCODE / AGC_AMD_ASYNC_BOOK / --no-prefetch say)
    archive_path = os.environ.get("AGC_BENCH_ARCHIVE", "") if rank == 0 or not single else ""
    if archive_path and world > 1 and not single:
        archive_path += f".{rank}"
    cmp_.create(archive_path, pack_card, K, None, SEG, MML, n_threads=threads)
    # reference preprocessing (once per archive, not timed): the reference's determine_splitters on the GPU
    t_spl0 = time.perf_counter()
    if args.positional_splitters:
        cmp_.set_splitters(synth_dev.positional_splitters(ref, off, K, SEG))
    else:
        cmp_.set_reference_dev(ref.data_ptr(), off)
    t_spl = time.perf_counter() - t_spl0
    # the reference genome is the first sample of every archive (src/app/main.cpp:106-114): it mints the
    # groups and their references.  Once per archive -> setup, not part of the per-sample hot path.
    t_ref0 = time.perf_counter()
    dc = None
    if single:
        from agc_amd.dist import DistCompressor
        dc = DistCompressor(cmp_, dist, rank, world, device=dev)

        def get_sample(i):
            """global sample i: 0 = the reference genome (rank 0), then one sample per rank and step, committed in rank order"""
            if i == 0:
                return "ref", names, ref.data_ptr(), off
            s_ = (i - rank) // world - (1 if rank == 0 else 0)
            return f"s{rank}_{s_}", names, packed_sample(s_)[0], sample_off[s_]

        dc.compress(1, get_sample)  # sample 0: minted on rank 0, its record (the whole reference set) broadcast
    else:
        cmp_.add_sample_dev("ref", names, ref.data_ptr(), off)
    cmp_.drain()  # the reference sample's entropy work (50 k reference streams) belongs to the setup, not to the timed steps
    t_ref = time.perf_counter() - t_ref0

    def add_step(s, tag):
        """one step = one sample per GPU; in single-archive mode the N samples of a step are committed in rank order"""
        if not single:
            # the next sample is known (as a reader that runs ahead of the compressor knows its next file): its
            # splitter scan is queued on the device beside this sample's classification / encode / registration.  Not across
            # the warm-up / timed boundary: every timed sample's scan runs inside the timed region.
            # FASTA bytes -> 2-bit layout: this sample's and the next one's packs were queued one and two steps ago (at the start of
            # the warm-up and of the timed region: here, and waited for); the one after next is queued now and runs beside this step
            packed_sample(s)
            nxt = s + 1 < n_steps and s + 1 != args.warmup
            if nxt:
                packed_sample(s + 1)
            if s + 2 < n_steps and (s + 2 < args.warmup) == (s < args.warmup):
                start_pack(s + 2)
            if nxt and not args.no_prefetch:
                cmp_.set_next_sample_packed_dev(samples[s + 1][0], sample_off[s + 1])
            cmp_.add_sample_packed_dev(f"{tag}{rank}_{s}", names, samples[s][0], sample_off[s])
            return
        # (each step: the N samples prepared in parallel.  AGC_BENCH_SERIAL_PREPARE=1, a measuring aid for ranks that SHARE one GPU:
        # every rank prepares at its own turn, so the per-stage times of config.single_archive_ms_per_sample_rank0 are those of an
        # uncontended device)
        dc.compress(1 + (s + 1) * world, get_sample, prefetch=not os.environ.get("AGC_BENCH_SERIAL_PREPARE"), start=1 + s * world)

    n_steps = args.steps + args.warmup
    # weak scaling: samples are partitioned round-robin over ranks (one archive shard per rank), no data-path collective
    # samples are RESIDENT IN HBM IN THE 2-BIT LAYOUT (0.25 B per base; include/agc_hip.h: agc_hip_packed): generated as codes,
    # packed, and the codes dropped.  Inside a step every kernel reads the packed words where they lie.
    from agc_amd import capi
    hctx = capi.Context.from_handle(cmp_.hip_ctx())
    # Default (round 5): a sample is resident as THE BYTES OF ITS FASTA FILE (header lines, --fasta-width letters per line, line
    # ends: 1 + 1 / width bytes per base) and the timed step makes the 2-bit layout from them (agc_hip_pack_fasta_begin / _end: the
    # reference's preprocess_raw_contig + the packing, one pass; its output buffers are allocated here, its work is timed).
    # --prepacked: packed before the timer starts, as in round 4.
    samples = [None] * n_steps      # (Packed, backing tensors) once packed
    sample_off = [off] * n_steps    # the contigs' symbol offsets as the pack returns them
    fasta = []                      # per sample: (raw bytes in HBM, n_raw, raw_begin, raw_end, (words, index, esc))
    pack_pending = {}
    pack_ms_each = []
    AS_BUILT_BPS["pack"] = 2.0 * (1.0 + 1.0 / args.fasta_width) + 0.25  # (both passes read the bytes)
    PACKED_BPS["pack"] = 1.0 + 1.0 / args.fasta_width + 0.25           # (one read + the words: the algorithmic figure)
    for s in range(n_steps):
        codes = synth_dev.make_sample(ref, tot, args.div, shard.sample_seed(1000, s, rank, world), dev)
        if args.prepacked:
            samples[s] = hctx.pack_dev(codes, tot)
        else:
            raw, n_raw, rb, re_ = synth_dev.make_fasta(codes, off, names, args.fasta_width)
            bufs = (torch.empty(int(hctx.L.agc_hip_packed_words_bytes(tot)) // 4 + 4, dtype=torch.int32, device=dev),
                    torch.empty(int(hctx.L.agc_hip_packed_index_bytes(tot)) // 4 + 1, dtype=torch.int32, device=dev),
                    torch.empty(64 * 1024, dtype=torch.uint8, device=dev))
            fasta.append((raw, n_raw, rb, re_, bufs))
        del codes
    torch.cuda.synchronize()

    def start_pack(s):
        # announced to the compressor, which queues the conversion where the sample call in progress leaves the GPU room for it
        # (behind its classification kernels, beside the registration's host work and the announced scan) -- or at once, in
        # packed_sample(), when no sample call comes in between (the first two samples of the warm-up and of the timed region)
        if samples[s] is None and s not in pack_pending:
            raw, n_raw, rb, re_, bufs = fasta[s]
            pack_pending[s] = cmp_.set_next_fasta_dev(raw, n_raw, rb, re_, bufs)

    def packed_sample(s):
        if samples[s] is None:
            start_pack(s)
            pk, keep, o_ = cmp_.finish_fasta_dev(pack_pending.pop(s))
            pack_ms_each.append(round(hctx.timing_get()["pack"][0], 3))  # (cumulative while timing is on: differences = each pack)
            if not np.array_equal(o_, np.asarray(off, np.uint64)):
                raise SystemExit(f"bench.py: the pack of sample {s} returned other contig offsets than the generator's")
            samples[s], sample_off[s] = (pk, keep), o_
        return samples[s]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        add_step(s, "w")
    cmp_.drain()
    st0 = cmp_.stats()
    cmp_.hip_timing(True)  # HIP events on the library's stream around every kernel of the timed region
    barrier()
    sec0 = dict(dc.seconds) if dc is not None else None
    bytes0 = (dc.bytes_broadcast, dc.bytes_p2p, dc.n_records) if dc is not None else None
    t0 = time.perf_counter()
    step_ms_each = []
    cg0 = cgroup_cpu_stat()
    for s in range(args.warmup, n_steps):
        ts_ = time.perf_counter()
        add_step(s, "s")
        step_ms_each.append(round((time.perf_counter() - ts_) * 1e3, 2))
    t_steps = time.perf_counter() - t0
    cg1 = cgroup_cpu_stat()
    sec1 = dict(dc.seconds) if dc is not None else None
    # (per TIMED sample: the reference sample's record -- the whole collection's references -- is setup and not averaged in)
    head_timed_mb = (dc.bytes_broadcast - bytes0[0]) / max(dc.n_records - bytes0[2], 1) / 1e6 if dc is not None else 0.0
    body_timed_mb = (dc.bytes_p2p - bytes0[1]) / max((dc.n_records - bytes0[2]) * (world - 1) // world, 1) / 1e6 if dc is not None else 0.0
    # Close(): zstd of every pending delta pack + metadata + footer -- the deferred part of the steps' work
    if single:
        dc.close(n_threads=threads)  # packs handed out to every rank's GPU, frames gathered by the writer
    else:
        cmp_.close(threads)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cg2 = cgroup_cpu_stat()
    tm = cmp_.hip_timing_get()
    st1 = cmp_.stats()
    stats = {k_: st1[k_] - st0[k_] for k_ in st1}
    if world > 1:
        t = torch.tensor([elapsed, t_steps], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_steps = [float(x) for x in t.tolist()]
        keys = ["bases", "segments", "lz_encoded", "delta_bytes", "middle_tried", "middle_split", "one_splitter", "new_groups", "zstd_in", "zstd_out"]
        agg = torch.tensor([stats[k_] for k_ in keys], device=red_dev, dtype=torch.float64)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        for k_, v in zip(keys, agg.tolist()):
            stats[k_] = v
    barrier()

    if rank == 0:
        value = stats["bases"] / elapsed / 1e9
        # Roofline (HBM-bound byte work).  Algorithmic bytes per kernel and step = the symbols the kernel was asked to look at
        # (host counters): scan = every symbol of the sample once; encode / estimate / cost vectors = every text once + its
        # reference once.  Two columns: the layout as built (`bytes_per_symbol` B per symbol) and SURVEY 8d's 2-bit figure
        # (0.25 B per symbol), which is what north_star's roofline target is quoted on.
        # Kernel time = HIP events on the library's own stream around every launch of the timed region, summed per step
        # (scan and encode are one launch per step; "costvec" = cost-vector parse + split-point reduction; "filter" = the
        # key-filter kernel that writes the "may match" bitmaps of the estimate / cost-vector parses: it reads every text once;
        # there is no expansion kernel any more: every row's time is that kernel's alone).
        n_rank_steps = max(args.steps * world, 1)
        sym = {"scan": stats["bases"], "pack": stats["bases"], "encode": stats["enc_text"] + stats["enc_ref"],
               "estimate": stats["est_text"] + stats["est_ref"], "costvec": stats["cv_text"] + stats["cv_ref"],
               "filter": stats["est_text"] + stats["cv_text"]}
        tab = pmc_table()
        kern = {}
        for name in ("pack", "scan", "encode", "estimate", "costvec", "filter"):
            ms_, n_ = tm[name]
            if not n_:
                continue
            ms_step = ms_ / max(args.steps, 1)
            sym_step = sym[name] / n_rank_steps
            row = {"ms_per_step": round(ms_step, 4), "launches_per_step": round(n_ / max(args.steps, 1), 2),
                   "symbols_per_step": int(sym_step)}
            for col, bps in (("as_built", AS_BUILT_BPS[name]), ("packed_2bit", PACKED_BPS[name])):
                ach = bps * sym_step / (ms_step * 1e-3) / 1e9
                row[col] = {"bytes_per_symbol": bps, "algorithmic_bytes": int(bps * sym_step), "achieved": round(ach, 2),
                            "frac": round(ach / HBM_PEAK_GBS, 5)}
            tr = pmc_traffic(tab, name)
            row["traffic"] = tr
            row["traffic_range"] = pmc_traffic_range(tab, name)
            row["waste"] = round(tr / row["as_built"]["algorithmic_bytes"], 2) if tr and row["as_built"]["algorithmic_bytes"] else None
            kern[name] = row
        # the entropy stage's kernel (zstd level 17 of the delta packs, one group of lanes per frame): its launches belong to
        # Close() (and to full packs during the steps); per "step" = its time / K like every other row.  Algorithmic bytes =
        # the packs it read + the frames it wrote.  It is a chain of dependent table look-ups per frame, not a streaming kernel:
        # the HBM fraction is the number the contract asks for, what bounds it is in DESIGN.md 4.6.
        ms_z, n_z = tm.get("zstd", (0.0, 0))
        if n_z and stats["zstd_dev_in"] > 0:
            dev_in = stats["zstd_dev_in"]
            dev_out = stats["zstd_dev_out"]  # (counted: the bytes of the frames the device wrote)
            alg = dev_in + dev_out
            ach = alg / (ms_z * 1e-3) / 1e9
            tr = pmc_traffic(tab, "zstd")
            kern["zstd"] = {"ms_per_step": round(ms_z / max(args.steps, 1), 4), "launches_per_step": round(n_z / max(args.steps, 1), 2),
                            "launches": int(n_z), "avg_launch_ms": round(ms_z / n_z, 3),
                            "bytes_in_per_launch": int(dev_in / n_z), "bytes_out_per_launch": int(dev_out / n_z),
                            "as_built": {"bytes_per_symbol": None, "algorithmic_bytes": int(alg / n_z), "achieved": round(ach, 3),
                                         "frac": round(ach / HBM_PEAK_GBS, 6)},
                            "packed_2bit": {"bytes_per_symbol": None, "algorithmic_bytes": int(alg / n_z), "achieved": round(ach, 3),
                                            "frac": round(ach / HBM_PEAK_GBS, 6)},
                            "traffic": tr, "traffic_range": pmc_traffic_range(tab, "zstd"),
                            "waste": round(tr / (alg / n_z), 2) if tr else None}
        dominant = max(kern, key=lambda k_: kern[k_]["ms_per_step"]) if kern else None
        dom = kern.get(dominant, {})
        per = lambda x: x / max(args.steps * world, 1)
        out = {
            "metric": "input Gbp/s compressed (create), hot path " + ("" if args.prepacked else "FASTA bytes in HBM -> 2-bit pack + ") + "scan + match + encode + zstd packing",
            "value": round(value, 3), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: GRCh38-shaped {args.gbp:g} Gbp reference (24 contigs), one {args.gbp:g} Gbp sample per GPU per step, "
                                   f"d={args.div:g}, k={K} l={MML} b={pack_card} s={SEG}"
                                   + (" (OVERLAP VARIANT: packs fill during the steps)" if pack_card != PACK else ""),
                       "input": ("samples packed into the 2-bit layout before the timer (--prepacked)" if args.prepacked else
                                 f"samples resident in HBM as the bytes of their FASTA files ({args.fasta_width} letters per line); every timed sample's "
                                 "conversion + 2-bit packing (agc_cmp_set_next_fasta_dev -> agc_hip_pack_fasta_*) runs inside the timed region, two samples ahead on its own stream, queued by the compressor"),
                       "stages_timed": "per step: " + ("" if args.prepacked else "FASTA bytes -> 2-bit words (one-pass kernel), ") + "splitter-scan kernel, hit fix-up, add_segment classification (one-splitter estimates and "
                                       "missing-middle split points on the GPU), group registration, index build of new references, LZ-diff "
                                       "encode kernel, delta D2H, pack bookkeeping, collection records (a sample's encode is collected and its "
                                       "bookkeeping done on a second thread beside the next sample's scan and classification; the next sample's "
                                       "scan is queued ahead); the entropy stage (zstd 17 of full delta packs: GPU kernel + host "
                                       "pool; 13/19 of new references: host pool) runs on a background "
                                       "thread beside the steps; after the last step: Close() = the same for every open pack + archive "
                                       "metadata, and the wait for it all.  Inputs resident in HBM; archive bytes produced, not written to disk.",
                       "setup_not_timed": f"determine_splitters ({'positional shortcut' if args.positional_splitters else 'GPU: enumerate + radix sort + singletons'}): "
                                          f"{t_spl:.2f} s; reference genome as first sample (mints ~{int(st0['new_groups'])} groups): {t_ref:.2f} s",
                       "steps_only_ms": round(t_steps / max(args.steps, 1) * 1e3, 3),
                       "step_ms_each_rank0": step_ms_each,
                       # the container's CPU quota at work (cgroup v2 cpu.stat deltas): CPU seconds used and time spent throttled
                       "cgroup_cpu": ({"steps": {"cpu_s": round((cg1.get("usage_usec", 0) - cg0.get("usage_usec", 0)) / 1e6, 3),
                                                 "throttled_s": round((cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e6, 3),
                                                 "nr_throttled": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0)},
                                       "close": {"cpu_s": round((cg2.get("usage_usec", 0) - cg1.get("usage_usec", 0)) / 1e6, 3),
                                                 "throttled_s": round((cg2.get("throttled_usec", 0) - cg1.get("throttled_usec", 0)) / 1e6, 3),
                                                 "nr_throttled": cg2.get("nr_throttled", 0) - cg1.get("nr_throttled", 0)}} if cg0 else None),
                       "pack_ms_cumulative_after_each_pack": pack_ms_each[-(args.steps + 2):],
                       **({"archive_sha256": _sha256_file(archive_path), "archive_bytes": os.path.getsize(archive_path)} if archive_path else {}),
                       "close_ms": round((elapsed - t_steps) * 1e3, 1),
                       "segments_per_step": int(per(stats["segments"])), "lz_encoded_per_step": int(per(stats["lz_encoded"])),
                       "missing_middle_per_step": int(per(stats["middle_tried"])), "one_splitter_per_step": int(per(stats["one_splitter"])),
                       "new_groups_per_step": int(per(stats["new_groups"])), "delta_bytes_per_step": int(per(stats["delta_bytes"])),
                       "zstd": {"version": cmp_.zstd_version(), "host_threads": threads, "in_bytes": int(stats["zstd_in"]), "out_bytes": int(stats["zstd_out"]),
                                "device_in_bytes": int(stats["zstd_dev_in"]), "device_call_s": round(stats["t_zstd_dev"], 3),
                                "host_pool_s": round(stats["t_zstd_host"], 3), "staging_s": round(stats["t_zstd_stage"], 3),
                                "stage_busy_s": round(stats["t_zstd"], 3), "caller_waited_s": round(stats["t_zstd_wait"], 3),
                                "overlap": round(1.0 - stats["t_zstd_wait"] / stats["t_zstd"], 3) if stats["t_zstd"] > 0 else None,
                                "mb_s_per_host_thread": round((stats["zstd_in"] - stats["zstd_dev_in"]) / 1e6 / max(stats["t_zstd_host"], 1e-9) / threads, 2)},
                       "host_stage_seconds_rank0": {k_: round(stats[k_], 4) for k_ in stats if k_.startswith("t_")},
                       # ONE archive from N ranks: rank 0's host milliseconds per sample of the timed steps -- prepare / commit of its
                       # own samples (prepare runs beside the other ranks'), head / body / apply of every sample's record (serial)
                       "single_archive_ms_per_sample_rank0": ({k_: round((sec1[k_] - sec0[k_]) * 1e3 / max(args.steps * (1 if k_ in ("prepare", "commit", "finish") else world), 1), 3)
                                                               for k_ in sec1} if sec1 is not None else None),
                       "parallelism": (f"samples round-robin over {world} GPUs into ONE archive: ordered commit, one {'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()} "
                                       f"broadcast per sample (64-byte message header + the commit record's head: ids, keys, new reference segments; "
                                       f"{dc.n_collectives} broadcasts for {dc.n_records} records), {head_timed_mb:.2f} MB per timed sample, "
                                       f"its delta body point to point to the writer ({body_timed_mb:.1f} MB per timed sample that is not the writer's own); "
                                       f"full packs are dealt to the ranks' entropy stages in the middle of the run ({dc.n_deals} deals, {dc.bytes_dealt / 1e6:.1f} MB of "
                                       "packs left the writer; a rank codes its share on a thread of its own beside its samples, the frames travel back at a "
                                       "later control step); at Close every rank receives its own byte range of the packs still open point to point, its GPU "
                                       "compresses it, rank 0 receives the frames as long as they are and writes") if single else
                                      f"samples round-robin over {world} GPU(s), one archive shard per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": KERNEL_SYMBOL.get(dominant), "achieved": dom.get("as_built", {}).get("achieved"),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom.get("as_built", {}).get("frac"),
                         "frac_packed_2bit": dom.get("packed_2bit", {}).get("frac"),
                         "traffic": dom.get("traffic"), "traffic_range": dom.get("traffic_range"), "waste": dom.get("waste"),
                         "traffic_source": (PMC_SUMMARY + " (max over dispatches, per launch; traffic_range = [FETCH_SIZE + WRITE_SIZE, 2 x FETCH_SIZE "
                                            "+ WRITE_SIZE]; traffic = the upper end for streaming kernels, the lower end for the zstd kernels, whose "
                                            "4-byte scattered loads the counter tallies as 64-B requests: profiles/r4/fetch_calibration.txt)") if dom.get("traffic") else None,
                         "algorithmic_bytes_per_launch": dom.get("as_built", {}).get("algorithmic_bytes"),
                         "avg_launch_ms": dom.get("avg_launch_ms", dom.get("ms_per_step")),
                         "dominant_by": "largest kernel time per step among ALL kernels of the path (scan, encode, estimate, cost vectors, "
                                        "key filter, zstd frames)",
                         "layout": "samples AND group references resident in HBM at 0.25 B per symbol (2-bit words + escaped blocks for "
                                   "anything outside ACGT); scan, key filter and the LZ parses read that layout where it lies, in either "
                                   "orientation: no expansion kernel, no reverse-complement staging (as_built == packed_2bit)",
                         "kernels": kern,
                         "kernel_ms_per_step_rank0": {n: round(v[0] / max(args.steps, 1), 4) for n, v in tm.items() if v[1]}},
        }
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.cpu_baseline_mbp)
            except Exception as e:  # the baseline is informational; never lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    cmp_.close_handle()  # tear the HIP context down before the interpreter (and any profiler) exits
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
