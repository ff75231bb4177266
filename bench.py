#!/usr/bin/env python3
"""bench.py -- AGC `create` hot path on MI355X: input Gbp/s compressed.

One "step" = one pass of the hot path (splitter scan -> add_segment classification ->
LZ-diff encode of every placed segment against its group reference -> pack bookkeeping)
over one synthetic human-scale sample that is already resident in HBM
(BASELINE.json configs[2]: GRCh38-shaped reference, 0.1 % divergence, k=31 l=15 b=100);
the zstd packing the steps defer (packs flush every b samples) runs in Close(), which is
inside the timed region.  The step is the same code path `agc_amd create` runs and whose
archives are byte-identical to the reference's (tests/test_gpu_archive.py).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` = bases of all ranks / max-over-ranks time.
roofline.* is for the dominant kernel (LZ encode), timed with HIP events on the library's
own stream; cpu_baseline is the reference CPU implementation (oracle/_ref/agc, when it was
prebuilt) or the oracle port, timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K, MML, SEG, PACK = 31, 15, 60000, 100
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gbp", type=float, default=3.0, help="bases per sample (Gbp)")
    ap.add_argument("--div", type=float, default=1e-3, help="per-base substitution rate")
    ap.add_argument("--cpu-baseline-mbp", type=float, default=200.0, help="size of the CPU baseline sample (Mbp per genome)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--positional-splitters", action="store_true",
                    help="skip determine_splitters: take the k-mer at every segment_size-th position (valid for an i.i.d. reference)")
    ap.add_argument("--threads", type=int, default=0, help="host threads for libzstd (default: all cores / n_gpus)")
    ap.add_argument("--single-archive", action="store_true",
                    help="N > 1: all ranks feed ONE archive (ordered commit from broadcast commit records, agc_amd/dist.py) instead of "
                         "one archive shard per rank; rank 0 writes and runs libzstd with all host threads")
    return ap.parse_args()


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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary of this workload
    (profiles/r1_final/pmc_summary.csv: separate --pmc FETCH_SIZE / WRITE_SIZE passes; values in KB;
    FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note).  None when the summary is absent."""
    fn = os.path.join(ROOT, "profiles", "r1_final", "pmc_summary.csv")
    try:
        fetch = write = None
        for line in open(fn):
            f = line.strip().split(",")
            if len(f) >= 5 and f[0].endswith(kernel.replace("lz_parse_kernel<ENCODE>", "lz_parse_kernel<0>")):
                if f[1] == "FETCH_SIZE":
                    fetch = float(f[4])   # max over dispatches = the full-size launches
                elif f[1] == "WRITE_SIZE":
                    write = float(f[4])
        if fetch is None:
            return None, None
        return int((2.0 * fetch + (write or 0.0)) * 1024), "profiles/r1_final/pmc_summary.csv (2 x FETCH_SIZE + WRITE_SIZE, per launch)"
    except OSError:
        return None, None


def cpu_baseline(args, mbp):
    """reference CPU implementation on a bounded twin of the workload (same generator, smaller):
    wall(create ref + 2 samples) - wall(create ref only), all host cores."""
    from agc_amd import synth
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "agc")
    rng = np.random.default_rng(12345)
    n = int(mbp * 1e6)
    ctg_len = [n // 4] * 4
    refc = [synth.random_seq(rng, l) for l in ctg_len]
    n_samples = 4
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

            def run(extra):
                t0 = time.time()
                subprocess.run(common + [os.path.join(td, "o.agc"), os.path.join(td, "ref.fa")] + extra,
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                return time.time() - t0
            t_ref = min(run([]), run([]))
            t_all = min(run(files), run(files))
            dt = max(t_all - t_ref, 1e-6)
            return {"value": n_samples * n / dt / 1e9, "unit": "Gbp/s", "cores": int(t_threads), "kind": "reference",
                    "sample": f"oracle/_ref/agc create -t {t_threads}: wall(ref + {n_samples} x {mbp:g} Mbp samples, d={args.div:g}) "
                              f"- wall(ref only) = {dt:.2f} s"}
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


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    from agc_amd import host, shard, synth_dev

    total = int(args.gbp * 1e9)
    ref, off = synth_dev.make_reference(total, 12345, dev)
    tot = int(off[-1])
    names = [f"chr{i + 1}" for i in range(len(off) - 1)]
    single = args.single_archive and world > 1
    threads = max(1, host_cpus() // world)
    if single:  # the writer rank does all the zstd work, the others need a few host threads only
        threads = max(1, host_cpus() - 2 * (world - 1)) if rank == 0 else 2
    if args.threads:
        threads = args.threads
    cmp_ = host.Compressor(local)
    if single:
        cmp_.set_distributed(rank, world, 0)
    # archive bytes are produced and discarded (out path ""): file I/O is not the path under test
    cmp_.create("", PACK, K, None, SEG, MML, n_threads=threads)
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
            return f"s{rank}_{s_}", names, samples[s_].data_ptr(), off

        dc.compress(1, get_sample)  # sample 0: minted on rank 0, its record (the whole reference set) broadcast
    else:
        cmp_.add_sample_dev("ref", names, ref.data_ptr(), off)
    t_ref = time.perf_counter() - t_ref0

    def add_step(s, tag):
        """one step = one sample per GPU; in single-archive mode the N samples of a step are committed in rank order"""
        if not single:
            cmp_.add_sample_dev(f"{tag}{rank}_{s}", names, samples[s].data_ptr(), off)
            return
        dc.compress(1 + (s + 1) * world, get_sample, start=1 + s * world)  # (each step: the N samples prepared in parallel)

    n_steps = args.steps + args.warmup
    # weak scaling: samples are partitioned round-robin over ranks (one archive shard per rank), no data-path collective
    samples = [synth_dev.make_sample(ref, tot, args.div, shard.sample_seed(1000, s, rank, world), dev) for s in range(n_steps)]
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        add_step(s, "w")
    st0 = cmp_.stats()
    cmp_.hip_timing(True)  # HIP events on the library's stream around every kernel of the timed region
    barrier()
    t0 = time.perf_counter()
    for s in range(args.warmup, n_steps):
        add_step(s, "s")
    t_steps = time.perf_counter() - t0
    # Close(): zstd of every pending delta pack + metadata + footer -- the deferred part of the steps' work
    cmp_.close(threads)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tm = cmp_.hip_timing_get()
    st1 = cmp_.stats()
    stats = {k_: st1[k_] - st0[k_] for k_ in st1}
    if world > 1:
        t = torch.tensor([elapsed, t_steps], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_steps = [float(x) for x in t.tolist()]
        keys = ["bases", "segments", "lz_encoded", "delta_bytes", "middle_tried", "middle_split", "one_splitter", "new_groups", "zstd_in", "zstd_out"]
        agg = torch.tensor([stats[k_] for k_ in keys], device=dev, dtype=torch.float64)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        for k_, v in zip(keys, agg.tolist()):
            stats[k_] = v
    barrier()

    if rank == 0:
        value = stats["bases"] / elapsed / 1e9
        # Roofline (HBM-bound byte work; SURVEY 8d algorithmic bytes, 1 B/symbol layout):
        #   scan   : every symbol of the sample read once                      -> 1.0 B/bp
        #   encode : text read once + matched reference read once              -> 2.0 B/bp
        # kernel time = HIP events on the library's own stream, averaged over the launches of the timed region.
        bases_per_launch = stats["bases"] / world / max(args.steps, 1)
        kern = {}
        for name, bpb in (("scan", 1.0), ("encode", 2.0)):
            ms_, n_ = tm[name]
            if n_:
                avg = ms_ / n_
                kern[name] = {"avg_launch_ms": round(avg, 4), "algorithmic_bytes_per_bp": bpb,
                              "achieved": round(bpb * bases_per_launch / (avg * 1e-3) / 1e9, 2)}
        dominant = max(kern, key=lambda k_: kern[k_]["avg_launch_ms"]) if kern else "encode"
        kname = {"scan": "scan_kernel", "encode": "lz_parse_kernel<ENCODE>"}[dominant]
        traffic, traffic_src = pmc_traffic(kname)
        per = lambda x: x / max(args.steps * world, 1)
        out = {
            "metric": "input Gbp/s compressed (create), hot path scan + match + encode + zstd packing",
            "value": round(value, 3), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: GRCh38-shaped {args.gbp:g} Gbp reference (24 contigs), one {args.gbp:g} Gbp sample per GPU per step, "
                                   f"d={args.div:g}, k={K} l={MML} b={PACK} s={SEG}",
                       "stages_timed": "per step: splitter-scan kernel, hit fix-up, add_segment classification (one-splitter estimates and "
                                       "missing-middle split points on the GPU), group registration, index build of new references, LZ-diff "
                                       "encode kernel, delta D2H, pack bookkeeping, collection records; after the last step: Close() = libzstd "
                                       "(level 17/13/19, host threads) of every pending pack + archive metadata.  Inputs resident in HBM; "
                                       "archive bytes produced, not written to disk.",
                       "setup_not_timed": f"determine_splitters ({'positional shortcut' if args.positional_splitters else 'GPU: enumerate + radix sort + singletons'}): "
                                          f"{t_spl:.2f} s; reference genome as first sample (mints ~{int(st0['new_groups'])} groups): {t_ref:.2f} s",
                       "steps_only_ms": round(t_steps / max(args.steps, 1) * 1e3, 3),
                       "async_encode": bool(os.environ.get("AGC_AMD_ASYNC_ENCODE")),
                       "close_ms": round((elapsed - t_steps) * 1e3, 1),
                       "segments_per_step": int(per(stats["segments"])), "lz_encoded_per_step": int(per(stats["lz_encoded"])),
                       "missing_middle_per_step": int(per(stats["middle_tried"])), "one_splitter_per_step": int(per(stats["one_splitter"])),
                       "new_groups_per_step": int(per(stats["new_groups"])), "delta_bytes_per_step": int(per(stats["delta_bytes"])),
                       "zstd": {"version": cmp_.zstd_version(), "host_threads": threads, "in_bytes": int(stats["zstd_in"]), "out_bytes": int(stats["zstd_out"])},
                       "host_stage_seconds_rank0": {k_: round(stats[k_], 4) for k_ in stats if k_.startswith("t_")},
                       "parallelism": (f"samples round-robin over {world} GPUs into ONE archive: ordered commit, one RCCL broadcast of the commit "
                                       f"record (new reference segments + deltas) per sample, {dc.bytes_broadcast / max(dc.next_sample, 1) / 1e6:.1f} MB each; "
                                       "rank 0 writes and runs libzstd") if single else
                                      f"samples round-robin over {world} GPU(s), one archive shard per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": kern.get(dominant, {}).get("achieved"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(kern.get(dominant, {}).get("achieved", 0.0) / HBM_PEAK_GBS, 5),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(kern.get(dominant, {}).get("algorithmic_bytes_per_bp", 0) * bases_per_launch),
                         "avg_launch_ms": kern.get(dominant, {}).get("avg_launch_ms"),
                         "dominant_by": "largest average launch time among the path's streaming kernels",
                         "kernels": {k_: dict(v, frac=round(v["achieved"] / HBM_PEAK_GBS, 5)) for k_, v in kern.items()},
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
