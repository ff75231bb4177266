"""Multi-GPU single-archive protocol (agc_amd/dist.py) on CPU: gloo, world size 2 and 3, every rank driving the
host pipeline through the CPU device stand-in (tests/devsim).  The one archive the writer rank produces must equal the
reference CLI's `create` output for the same files (recorded sha256), i.e. the ordered commit of samples classified on
different ranks -- with the newly minted references travelling inside the broadcast records -- loses nothing."""
import ctypes as C
import hashlib
import json
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import collections as COLL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "archives.json")))

from agc_amd.fasta import read_codes as fasta_codes  # noqa: E402


def _sim_zstd_batch():
    """(packs back to back, offsets) -> (level-17 frames back to back, offsets) through the CPU stand-in's agc_hip_zstd17_batch"""
    from tests.devsim import build as simbuild
    sim = C.CDLL(simbuild.SIM_HIP)
    sim.agc_hip_zstd17_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]

    def run(src, off):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        if src.size == 0:
            src = np.zeros(1, np.uint8)
        cap = int(off[-1]) + 32 * n + 64
        dst = np.zeros(cap, np.uint8)
        doff = np.zeros(n + 1, np.uint64)
        assert sim.agc_hip_zstd17_batch(C.c_void_p(1), n, src.ctypes.data, off.ctypes.data, dst.ctypes.data, cap, doff.ctypes.data) == 0
        return dst[:int(doff[-1])], doff
    return run


def _worker(rank, world, port, name, files, out_path, q, on_gpu=False, prefetch=False, append_to=None, backend="gloo", cmp_world=None):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if backend == "nccl":  # RCCL: one process per GPU, the communicator bound to the device at once
            import torch
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from agc_amd import host
        from agc_amd.dist import DistCompressor
        device = None
        if on_gpu:  # the product libraries and the real kernels; samples and records live in HBM
            import torch
            lib = host.load()
            device = torch.device("cuda:0")
        else:       # host pipeline on the device stand-in (never the product library here)
            from tests.devsim import build as simbuild
            lib = host.bind(C.CDLL(simbuild.SIM_HOST))
        args = list(name) if isinstance(name, (list, tuple)) else COLL.CONFIGS[name][0]  # (a list: the CLI options themselves -- scripts/fuzz_deals.py)
        opt = {"-k": 31, "-l": 20, "-s": 60000, "-b": 50}
        for i in range(0, len(args) - 1):
            if args[i] in opt:
                opt[args[i]] = int(args[i + 1])
        cmp_ = host.Compressor(lib=lib)
        # (cmp_world: what the compressor is told -- a lone rank that is told of two still makes commit records and hands its packs
        # out at Close, so the record transport runs without a second process)
        cmp_.set_distributed(rank, cmp_world or world, 0)
        if append_to is not None:  # every rank loads the input archive; the writer copies it into the new one
            cmp_.append(append_to, out_path if rank == 0 else "", concatenated="-c" in args, adaptive="-a" in args, n_threads=2)
        else:
            cmp_.create(out_path if rank == 0 else "", pack_cardinality=opt["-b"], k=opt["-k"], ref_file=files[0], segment_size=opt["-s"],
                        min_match_len=opt["-l"], concatenated="-c" in args, adaptive="-a" in args, n_threads=2)
        # (the stand-in's encoder from the start: full packs are dealt to the ranks in the middle of the run, not only at Close)
        dc = DistCompressor(cmp_, dist, rank, world, device=device, zstd_raw=None if on_gpu else _sim_zstd_batch())
        keep = {}
        units = None
        if "-c" in args:  # the reference's registration units: runs of -b contigs across the files, sample name ""
            from agc_amd.dist import archive_contig_names, concatenated_units
            n0, names0, b0 = archive_contig_names(append_to) if append_to is not None else (0, (), None)
            units = concatenated_units([fasta_codes(f)[0] for f in files], b0 or opt["-b"], already=n0, seen=names0)

        def get_sample(i):
            if units is not None:
                names, parts, off = [], [], [0]
                for fi, ci in units[i]:
                    fn, fc, fo = fasta_codes(files[fi])
                    names.append(fn[ci])
                    parts.append(fc[int(fo[ci]):int(fo[ci + 1])])
                    off.append(off[-1] + parts[-1].size)
                codes = np.concatenate(parts + [np.full(4096, 4, np.uint8)])
                sn, off = "", np.asarray(off, np.uint64)
            else:
                names, codes, off = fasta_codes(files[i])
                sn = os.path.basename(files[i])
            for suf in (".gz", ".fa", ".fasta", ".fna"):
                sn = sn[:-len(suf)] if sn.endswith(suf) else sn
            if on_gpu:
                keep[i] = torch.from_numpy(np.concatenate([codes, np.full(4096, 4, np.uint8)])).to(device)
                torch.cuda.synchronize()
                return sn, names, keep[i].data_ptr(), off
            keep[i] = codes
            return sn, names, codes.ctypes.data, off

        if prefetch or units is not None:
            dc.compress(len(files) if units is None else len(units), get_sample, prefetch=prefetch)  # (also in adaptive mode: a prepare that needs new splitters waits for its turn)
        for i, f in enumerate(files if not prefetch and units is None else []):
            if dc.owner_of(i) == rank:
                names, codes, off = fasta_codes(f)
                sn = os.path.basename(f)
                for suf in (".gz", ".fa", ".fasta", ".fna"):
                    sn = sn[:-len(suf)] if sn.endswith(suf) else sn
                if on_gpu:
                    d_codes = torch.from_numpy(np.concatenate([codes, np.full(4096, 4, np.uint8)])).to(device)
                    torch.cuda.synchronize()
                    dc.add_sample(sn, names, d_codes.data_ptr(), off)
                else:
                    dc.add_sample(sn, names, codes.ctypes.data, off)
            else:
                dc.add_sample()
        if on_gpu:
            dc.close()
        else:
            dc.close(zstd_raw=_sim_zstd_batch())  # the stand-in's agc_hip_zstd17_batch (same encoder headers as the kernel)
        st = cmp_.stats()
        cmp_.close_handle()
        q.put((rank, "ok", dc.bytes_broadcast, st["new_groups"], st["revalidated"], st["reprepared"], sum(1 for i in range(len(files)) if i % world == rank),
               dc.n_deals, dc.bytes_dealt))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error: %r" % (e,), 0, 0, 0, 0, 0, 0, 0))


def _run(name, world, tmp_path, on_gpu, prefetch=False, files=None, append_to=None, want=None):
    files = COLL.build(name, str(tmp_path / "in")) if files is None else files
    out = str(tmp_path / "dist.agc")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, name, files, out, q, on_gpu, prefetch, append_to)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=300) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert all(r[1] == "ok" for r in res), res
    got = open(out, "rb").read()
    want = GOLD[name] if want is None else want
    assert len(got) == want["size"]
    assert hashlib.sha256(got).hexdigest() == want["sha256"]
    # every rank saw every record and minted the same groups
    assert len({r[2] for r in res}) == 1 and res[0][2] > 0
    assert len({r[3] for r in res}) == 1
    return res


@pytest.mark.parametrize("name,world", [("syn_snp", 2), ("syn_mixed", 2), ("syn_viral", 3), ("syn_shuffled", 2), ("syn_adaptive", 2),
                                        ("syn_adaptive", 3), ("toy_c1", 2),
                                        # the twins of BASELINE configs[2] / [3] / [4] (exact parameters, 1/100 size)
                                        # (world 2 and 3 between this test and the prefetching one below: the big twins take 30-40 s each on the CPU stand-in)
                                        ("syn_c3_twin", 2), ("syn_c4_twin", 3), ("syn_c5_twin", 2), ("syn_c5_twin", 3)])
def test_one_archive_from_n_ranks_equals_the_reference(name, world, tmp_path):
    from tests.devsim import build as simbuild
    simbuild.build()
    _run(name, world, tmp_path, on_gpu=False)


@pytest.mark.parametrize("name,world", [("syn_snp", 2), ("syn_snp", 3), ("syn_mixed", 2), ("syn_mixed", 3), ("syn_shuffled", 3), ("syn_viral", 2),
                                        ("syn_adaptive", 2), ("syn_adaptive", 3), ("syn_c3_twin", 3), ("syn_c5_twin", 2), ("syn_c5_twin", 3)])
def test_prefetching_ranks_still_write_the_reference_archive(name, world, tmp_path):
    """every rank classifies and speculatively encodes its next sample BEFORE the samples in front of it are committed; at its
    turn only the decisions that read changed state are revalidated -- the archive must not notice"""
    from tests.devsim import build as simbuild
    simbuild.build()
    res = _run(name, world, tmp_path, on_gpu=False, prefetch=True)
    if name == "syn_snp":  # groups are minted by every sample here: some prepared decisions must have been taken again
        assert sum(r[4] for r in res) > 0
    if name in ("syn_adaptive", "syn_c5_twin"):
        # adaptive mode: a prepare ahead of the turn must not extend the splitter set -- the samples that bring new splitters (or wait
        # while others do) were prepared again at their turn, the others kept their speculative prepare
        again, own = sum(r[5] for r in res), sum(r[6] for r in res)
        assert 0 < again < own, (again, own)
    else:
        assert sum(r[5] for r in res) == 0


@pytest.mark.parametrize("name,world,every", [("syn_viral", 2, 1), ("syn_viral", 3, 2), ("syn_snp", 3, 1), ("syn_mixed", 2, 1), ("syn_adaptive", 2, 1),
                                              ("syn_adaptive", 3, 3), ("syn_viral_c", 2, 1)])
def test_full_packs_are_dealt_to_the_ranks_in_the_middle_of_the_run(name, world, every, tmp_path, monkeypatch):
    """the reference's workers code a delta pack the moment it is full while the others go on (segment.cpp:34-80); here the writer deals
    the full packs to every rank's entropy stage a few samples after they filled (AGC_AMD_DEAL_MIN_MB=0: as soon as there is one;
    AGC_AMD_DEAL_EVERY: control step every that many samples), the ranks code their shares on threads of their own while the samples
    go on, the frames come back at later control steps -- and Close only deals what is still open.  Same archive; deals did happen."""
    from tests.devsim import build as simbuild
    simbuild.build()
    monkeypatch.setenv("AGC_AMD_DEAL_MIN_MB", "0")
    monkeypatch.setenv("AGC_AMD_DEAL_EVERY", str(every))
    res = _run(name, world, tmp_path, on_gpu=False, prefetch=True)
    assert len({r[7] for r in res}) == 1, [r[7] for r in res]   # every rank saw the same number of deals
    if not name.startswith("syn_adaptive"):  # (7 samples, one fill at the fifth, parked by the bookkeeping thread: Close may come first)
        assert res[0][7] > 0
    if name in ("syn_snp", "syn_mixed"):  # (a fill of the 30 kb genomes is one pack: the writer's own share; syn_adaptive: see above)
        assert sum(r[8] for r in res) > 0                                          # ... and pack bytes did leave the writer


@pytest.mark.parametrize("name,world,cap0,cap_max", [("syn_mixed", 2, 128, 4096), ("syn_snp", 3, 4096, 1 << 20), ("syn_adaptive", 2, 72, 72),
                                                     ("syn_c5_twin", 3, 1000, 30000)])
def test_record_heads_longer_than_the_message_take_a_second_broadcast(name, world, cap0, cap_max, tmp_path, monkeypatch):
    """the head travels as ONE message (64-byte header + head) whose capacity every rank derives from the sizes seen so far; a head
    that does not fit announces its size and sends its rest in a second broadcast.  With the capacities shrunk (72 bytes: room for
    the header and 8 bytes of head) every path of that is walked through, prefetching ranks included; same archive."""
    from tests.devsim import build as simbuild
    simbuild.build()
    monkeypatch.setenv("AGC_AMD_DIST_MSG_CAP0", str(cap0))
    monkeypatch.setenv("AGC_AMD_DIST_MSG_CAP_MAX", str(cap_max))
    _run(name, world, tmp_path, on_gpu=False, prefetch=True)


@pytest.mark.parametrize("name,world,prefetch", [("syn_viral_c", 2, False), ("syn_viral_c", 3, True), ("syn_adaptive_c", 2, True)])
def test_concatenated_mode_from_n_ranks_equals_the_reference(name, world, prefetch, tmp_path):
    """-c: the units dealt round-robin are the reference's registration units (runs of -b contigs across the files, every contig
    a sample of its own, then the empty registration the reference sends at the end) -- agc_amd.dist.concatenated_units"""
    from tests.devsim import build as simbuild
    simbuild.build()
    _run(name, world, tmp_path, on_gpu=False, prefetch=prefetch)


@pytest.mark.parametrize("plan,world", [("snp_4_3", 2), ("mixed_3_3", 2), ("viral_25_15", 3), ("shuffled_2_4", 2), ("adaptive_3_4", 2),
                                        ("viral_c_1_1", 2), ("adaptive_c_4_3", 2)])  # (the last two: append together with -c)
def test_append_from_n_ranks_equals_the_reference(plan, world, tmp_path):
    """`append` in the N-rank mode: every rank loads the input archive (groups packed, references decoded into its own HBM when a
    record first adds to a group -- apply_record unpacks exactly where the owner's registration does), samples are prepared at
    their turn (a packed group answers Estimate with 0 until it is unpacked: nothing can be classified ahead), the writer copies
    the old parts and appends.  Step 0 (`create`) is made by the CLI; the appended archive must be the reference CLI's."""
    from tests.devsim import build as simbuild
    cli = simbuild.build()
    coll, steps = COLL.APPEND_PLANS[plan]
    args, _ = COLL.CONFIGS[coll]
    files = COLL.build(coll, str(tmp_path / "in"))
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "archives_append.json")))[plan]
    import subprocess
    step0 = str(tmp_path / "step0.agc")
    subprocess.run([cli, "create"] + args + ["-t", "4", "-o", step0] + files[:steps[0]], check=True, capture_output=True, timeout=300)
    assert hashlib.sha256(open(step0, "rb").read()).hexdigest() == gold[0]["sha256"]
    _run(coll, world, tmp_path, on_gpu=False, files=files[steps[0]:steps[0] + steps[1]], append_to=step0, want=gold[1])


def _one_rank(name, tmp_path, on_gpu, backend):
    files = COLL.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "dist.agc")
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_worker, args=(0, 1, port, name, files, out, q, on_gpu, True, None, backend, 2))
    p_.start()
    p_.join(timeout=600)
    assert not p_.is_alive()
    res = q.get(timeout=10)
    assert res[1] == "ok", res
    assert res[2] > 0  # (bytes of record heads that went through the broadcast)
    got = open(out, "rb").read()
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("name", ["syn_mixed", "syn_adaptive"])
def test_one_rank_told_of_two_makes_records_and_deals_its_packs(name, tmp_path):
    """the harness of the RCCL test below on the CPU (gloo, stand-in): one process whose compressor is told it is the writer of two ranks
    makes every commit record, publishes it to a world of one and codes its own share at Close; same archive"""
    from tests.devsim import build as simbuild
    simbuild.build()
    _one_rank(name, tmp_path, False, "gloo")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["syn_mixed", "syn_c3_twin"])
def test_record_transport_under_rccl_with_one_rank(name, tmp_path):
    """RCCL readiness on a one-GPU box: ONE process, backend nccl (= RCCL) bound to cuda:0, the compressor told it is the writer of a
    two-rank job -- every sample's commit record is made, copied pinned -> HBM into the device message buffer (`_dmsg`), broadcast
    (a one-rank collective on HBM tensors), the RCCL placement check (all_gather of device indices) and the warm-up run, Close deals the
    packs through the collect / provide path and this rank's GPU codes its share from HBM.  What a second rank would add -- send /
    recv of bodies and packs -- is not reachable with one process; the archive must still be the reference's."""
    from agc_amd import build
    build.build_host()
    _one_rank(name, tmp_path, True, "nccl")


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("syn_snp", 2), ("syn_mixed", 3)])
def test_prefetching_ranks_on_the_gpu(name, world, tmp_path):
    from agc_amd import build
    build.build_host()
    _run(name, world, tmp_path, on_gpu=True, prefetch=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("syn_snp", 2), ("syn_adaptive", 2), ("syn_mixed", 3), ("syn_c3_twin", 2), ("syn_c4_twin", 3), ("syn_c5_twin", 2)])
def test_one_archive_from_n_ranks_on_the_gpu(name, world, tmp_path):
    """the same protocol with the product libraries: N processes share cuda:0 (gloo moves the HBM-resident records; on a multi-GPU
    node the backend is nccl = RCCL, see bench.py --single-archive), real kernels on every rank, references registered from the
    HBM copy of the record"""
    from agc_amd import build
    build.build_host()
    _run(name, world, tmp_path, on_gpu=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("syn_snp", 2), ("syn_mixed", 3), ("syn_viral", 2)])
def test_full_packs_dealt_in_the_middle_of_the_run_on_the_gpu(name, world, tmp_path, monkeypatch):
    """the deals of test_full_packs_are_dealt_to_the_ranks_in_the_middle_of_the_run with the product libraries: a rank's share is coded by
    its GPU's entropy stream (agc_hip_zstd17_batch_dev on a thread of its own) WHILE the same context prepares and commits the next
    samples on the steps' streams; the frames travel back at a later control step.  Same archive."""
    from agc_amd import build
    build.build_host()
    monkeypatch.setenv("AGC_AMD_DEAL_MIN_MB", "0")
    monkeypatch.setenv("AGC_AMD_DEAL_EVERY", "1")
    res = _run(name, world, tmp_path, on_gpu=True, prefetch=True)
    assert len({r[7] for r in res}) == 1 and res[0][7] > 0, [r[7] for r in res]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["syn_mixed", "syn_viral_c", "append:mixed_3_3", "append:viral_c_1_1"])
def test_dist_create_front_end_on_the_gpu(name, tmp_path):
    """python -m agc_amd.dist_create under torch.distributed.run, two ranks sharing cuda:0 (gloo): the user-facing multi-GPU create
    (syn_viral_c: -c, the registration units of the concatenated mode dealt over the ranks; append:<plan>: --append onto the archive
    the single-GPU CLI made of the plan's first step)"""
    import subprocess
    import sys
    from agc_amd import build
    build.build_host()
    extra, want = [], None
    if name.startswith("append:"):
        plan = name.split(":")[1]
        name, steps = COLL.APPEND_PLANS[plan]
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "archives_append.json")))[plan]
        want = gold[1]["sha256"]
    args, _ = COLL.CONFIGS[name]
    files = COLL.build(name, str(tmp_path / "in"))
    if want is not None:
        step0 = str(tmp_path / "step0.agc")
        subprocess.run([build.HOST_BIN, "create"] + args + ["-t", "4", "-o", step0] + files[:steps[0]], check=True, capture_output=True, timeout=300)
        assert hashlib.sha256(open(step0, "rb").read()).hexdigest() == gold[0]["sha256"]
        extra, files = ["--append", step0], files[steps[0]:steps[0] + steps[1]]
    else:
        want = GOLD[name]["sha256"]
    out = str(tmp_path / "d.agc")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "agc_amd.dist_create", "--backend", "gloo"] + extra + args + ["-t", "4", "-o", out] + files
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert os.path.exists(out), r.stderr[-3000:]
    got = open(out, "rb").read()
    assert hashlib.sha256(got).hexdigest() == want, r.stderr[-2000:]


# ---- edge cases of the prepare / commit split ------------------------------------------------------------
def _edge_samples():
    """splitters chosen by hand (SetSplitters): three inside a genome G, one in the middle of an unrelated sequence Z.
    sample 1 mints only ONE-SIDED groups -- (kZ, none) and (none, kZ): its contig Z holds a single splitter that has no
    terminator yet -- while sample 2, prepared on the other rank before sample 1 is committed, carries a mutated Z whose
    segments must join exactly those groups.  Sample 3 has no contigs at all."""
    from agc_amd import synth
    from oracle import agc_oracle as O
    rng = np.random.default_rng(99)
    k = 21
    G = synth.random_seq(rng, 12_000)
    Z = synth.random_seq(rng, 5_000)

    def can(seq, end):  # canonical k-mer ending at `end`, left-aligned as CKmer keeps it
        s = seq[end - k + 1:end + 1].astype(np.uint64)
        d = np.uint64(0)
        r = np.uint64(0)
        for j in range(k):
            d = (d << np.uint64(2)) | s[j]
            r = (r << np.uint64(2)) | (np.uint64(3) - s[k - 1 - j])
        sh = np.uint64(64 - 2 * k)
        return min(int(d << sh), int(r << sh))

    spl = np.array(sorted({can(G, 3000), can(G, 6000), can(G, 9000), can(Z, 2500)}), np.uint64)
    samples = [
        ("s0", ["g"], [G]),
        ("s1", ["g", "z"], [synth.mutate(rng, G, 0.003), Z]),
        ("s2", ["g", "z"], [synth.mutate(rng, G, 0.003), synth.mutate(rng, Z, 0.003)]),
        ("s3", [], []),
        ("s4", ["z", "g"], [synth.mutate(rng, Z, 0.003), synth.mutate(rng, G, 0.003)]),
    ]
    return k, spl, samples


def _edge_worker(rank, world, port, out_path, q, prefetch):
    try:
        from agc_amd import host
        from tests.devsim import build as simbuild
        lib = host.bind(C.CDLL(simbuild.SIM_HOST))
        k, spl, samples = _edge_samples()
        cmp_ = host.Compressor(lib=lib)
        if world > 1:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(port)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            cmp_.set_distributed(rank, world, 0)
        cmp_.create(out_path if rank == 0 else "", pack_cardinality=3, k=k, ref_file=None, segment_size=1000, min_match_len=18, n_threads=2)
        cmp_.set_splitters(spl)
        keep = {}

        def get_sample(i):
            name, names, ctgs = samples[i]
            off = np.zeros(len(ctgs) + 1, np.uint64)
            off[1:] = np.cumsum([c.size for c in ctgs])
            keep[i] = np.concatenate(ctgs + [np.full(64, 4, np.uint8)])
            return name, names, keep[i].ctypes.data, off

        if world > 1:
            from agc_amd.dist import DistCompressor
            dc = DistCompressor(cmp_, dist, rank, world, device=None)
            dc.compress(len(samples), get_sample, prefetch=prefetch)
        else:
            for i in range(len(samples)):
                cmp_.add_sample_dev(*get_sample(i))
        if world > 1:
            dc.close(zstd_raw=_sim_zstd_batch())
        else:
            cmp_.close()
        st = cmp_.stats()
        cmp_.close_handle()
        q.put((rank, "ok", st["new_groups"]))
        if world > 1:
            dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error: %r" % (e,), 0))


def _edge_run(world, tmp_path, prefetch, tag):
    out = str(tmp_path / f"edge_{tag}.agc")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_edge_worker, args=(r, world, port, out, q, prefetch)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=300) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert all(r[1] == "ok" for r in res), res
    assert len({r[2] for r in res}) == 1, res  # every rank minted the same number of groups
    return open(out, "rb").read(), res[0][2]


def _failing_owner_worker(rank, world, port, out_path, q, prefetch, owner_side=False):
    """as _edge_worker, but the sample rank 1 owns carries the name of the sample before it: the collection refuses it
    (collection_v3.cpp:682-708) and the commit fails on its owner"""
    try:
        from agc_amd import host
        from agc_amd.dist import DistCompressor
        from tests.devsim import build as simbuild
        lib = host.bind(C.CDLL(simbuild.SIM_HOST))
        k, spl, samples = _edge_samples()
        if not owner_side:
            samples[1] = (samples[0][0], samples[1][1], samples[1][2])
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=120))
        cmp_ = host.Compressor(lib=lib)
        cmp_.set_distributed(rank, world, 0)
        cmp_.create(out_path if rank == 0 else "", pack_cardinality=3, k=k, ref_file=None, segment_size=1000, min_match_len=18, n_threads=2)
        cmp_.set_splitters(spl)
        keep = {}

        def get_sample(i):
            name, names, ctgs = samples[i]
            off = np.zeros(len(ctgs) + 1, np.uint64)
            off[1:] = np.cumsum([c.size for c in ctgs])
            keep[i] = np.concatenate(ctgs + [np.full(64, 4, np.uint8)])
            return name, names, keep[i].ctypes.data, off

        dc = DistCompressor(cmp_, dist, rank, world, device=None)
        t0 = __import__("time").perf_counter()
        try:
            if owner_side:  # the commit itself fails on the owner of sample 1: every rank goes through add_sample
                for i in range(len(samples)):
                    if i == 1 and rank == dc.owner_of(1):
                        def boom(*a, **k):
                            raise RuntimeError("the owner's commit failed")
                        cmp_.add_sample_dev = boom
                    dc.add_sample(*get_sample(i)) if rank == dc.owner_of(i) else dc.add_sample()
            else:
                dc.compress(len(samples), get_sample, prefetch=prefetch)
            q.put((rank, "no error", 0.0))
        except RuntimeError as e:
            q.put((rank, "raised: %s" % (e,), __import__("time").perf_counter() - t0))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "error: %r" % (e,), 0.0))


@pytest.mark.parametrize("prefetch", [False, True])
def test_a_sample_that_cannot_be_committed_stops_every_rank_at_once(prefetch, tmp_path):
    """a sample that cannot be committed (here: a sample name the collection holds already) must stop every rank at once,
    not leave the others waiting in a broadcast until the collective times out"""
    from tests.devsim import build as simbuild
    simbuild.build()
    out = str(tmp_path / "never.agc")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_failing_owner_worker, args=(r, 2, port, out, q, prefetch)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=200) for _ in ps)
    [p.join(timeout=30) for p in ps]
    [p.kill() for p in ps if p.is_alive()]
    # (the writer refuses the record when it applies it -- the collection's own check -- and says why on stderr; the owner learns
    # of it at its next exchange with the writer.  A failure on the owner's side of the commit goes out as an "AGCX" header in the
    # record's place: DistCompressor._announce_failure)
    assert all(r[1].startswith("raised:") for r in res), res
    assert "ApplyRecord failed" in res[0][1] or "failed to commit" in res[0][1], res
    assert max(r[2] for r in res) < 60, res  # (at once, not at the collective's timeout)


def test_a_commit_that_fails_on_its_owner_is_announced_in_the_record_s_place(tmp_path):
    """the owner's commit raises before there is a record: the ranks waiting in that record's broadcast get an "AGCX" header and raise"""
    from tests.devsim import build as simbuild
    simbuild.build()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_failing_owner_worker, args=(r, 2, port, str(tmp_path / "never.agc"), q, False, True)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=200) for _ in ps)
    [p.join(timeout=30) for p in ps]
    [p.kill() for p in ps if p.is_alive()]
    assert "the owner's commit failed" in res[1][1], res
    assert "failed to commit its sample" in res[0][1], res
    assert max(r[2] for r in res) < 60, res


@pytest.mark.parametrize("world", [2, 3])
def test_one_sided_groups_minted_between_prepare_and_commit_and_an_empty_sample(world, tmp_path):
    """a group keyed (k-mer, none) minted by a sample committed between another rank's PrepareSampleDevice and CommitPrepared
    must be joined, not minted twice; a sample without contigs must not derail the record stream"""
    from tests.devsim import build as simbuild
    simbuild.build()
    want, n_groups = _edge_run(1, tmp_path, False, "single")
    got, n2 = _edge_run(world, tmp_path, True, f"w{world}")
    assert n2 == n_groups
    assert got == want


def test_concatenated_units_continue_the_input_archive():
    """-c mode, pure bookkeeping: units of -b contigs across the files, duplicates skipped, the closing (possibly empty) unit; with an
    input archive (append) the first unit completes the batch it ended in and its contigs are not taken again
    (agc_compressor.cpp:2150-2153, 2201-2205)"""
    from agc_amd.dist import concatenated_units
    names = [["a", "b", "c"], ["d", "b", "e", "f"], ["g"]]
    assert concatenated_units(names, 3) == [[(0, 0), (0, 1), (0, 2)], [(1, 0), (1, 2), (1, 3)], [(2, 0)]]
    assert concatenated_units(names, 7) == [[(0, 0), (0, 1), (0, 2), (1, 0), (1, 2), (1, 3), (2, 0)], []]
    # five samples in the archive, -b 3: two more complete its second batch; "c" is in the archive already
    assert concatenated_units(names, 3, already=5, seen=["c", "x"]) == [[(0, 0)], [(0, 1), (1, 0), (1, 2)], [(1, 3), (2, 0)]]
    assert concatenated_units([[]], 4, already=8) == [[]]
