"""Host logic of the create path on CPU: the unchanged host sources (agc_amd/csrc/host/) linked against the CPU device
stand-in of tests/devsim/ (include/agc_hip.h on top of the oracle) must write the reference's archives byte for byte.
This covers registration order, group/pack bookkeeping, adaptive mode, -c mode, zstd streams, collection metadata and
the container without a GPU; the HIP kernels themselves are covered by the `-m gpu` tests against the same goldens."""
import hashlib
import json
import os
import subprocess

import pytest

from tests import collections as C
from tests.devsim import build as simbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "archives.json")))


@pytest.fixture(scope="module")
def cli():
    return simbuild.build()


def _create(cli, args, files, out, threads="4"):
    r = subprocess.run([cli, "create"] + args + ["-t", threads, "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr[-2000:]
    return open(out, "rb").read()


@pytest.mark.parametrize("name", list(C.CONFIGS))
def test_host_pipeline_writes_the_reference_archive(cli, name, tmp_path):
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert len(got) == GOLD[name]["size"]
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


GOLD_APPEND = json.load(open(os.path.join(ROOT, "tests", "golden", "archives_append.json")))


@pytest.mark.parametrize("plan", list(C.APPEND_PLANS))
def test_append_writes_the_reference_archive(cli, plan, tmp_path):
    """create + append sequences: every intermediate and final archive equals the reference CLI's (recorded sha256),
    including its append-only behaviour for groups that are still packed (Estimate = 0, empty cost vectors)"""
    got = C.run_append_plan(cli, plan, str(tmp_path))
    want = GOLD_APPEND[plan]
    assert [len(x) for x in got] == [w["size"] for w in want]
    assert [hashlib.sha256(x).hexdigest() for x in got] == [w["sha256"] for w in want]


def test_append_round_trips_through_the_reader(cli, tmp_path):
    from agc_amd import build, reader
    from tests.test_read_cpu import fasta_text, parse_fasta
    build.build_read()
    C.run_append_plan(cli, "adaptive_1_2_4", str(tmp_path))
    a = reader.CAGCFile()
    assert a.Open(str(tmp_path / "step2.agc"))
    files = C.build("syn_adaptive", str(tmp_path / "in"))
    assert a.NSample() == len(files)
    for f in files:
        sn = os.path.basename(f)[:-3]
        assert a.GetSampleFasta(sn) == fasta_text(parse_fasta(f)), sn


def test_append_rejects_samples_already_present(cli, tmp_path):
    args, _ = C.CONFIGS["syn_mixed"]
    files = C.build("syn_mixed", str(tmp_path / "in"))
    base = str(tmp_path / "base.agc")
    _create(cli, args, files[:3], base)
    out = str(tmp_path / "o.agc")
    r = subprocess.run([cli, "append", "-o", out, base, files[1], files[4]], capture_output=True, text=True, timeout=300)
    assert "is already in the archive" in r.stderr and r.returncode == 0
    from agc_amd import build, reader
    build.build_read()
    a = reader.CAGCFile()
    assert a.Open(out) and a.NSample() == 4 and "m3" in a.ListSample()


def test_append_to_missing_archive_reports_and_exits_zero(cli, tmp_path):
    r = subprocess.run([cli, "append", "-o", str(tmp_path / "o.agc"), str(tmp_path / "none.agc"), __file__], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Cannot open archive" in r.stderr


@pytest.mark.parametrize("name", ["toy_c1", "syn_snp", "syn_mixed", "syn_viral", "syn_viral_c", "syn_adaptive_c", "syn_c5_twin"])
def test_host_pipeline_with_every_pack_on_the_device_entropy_stage(cli, name, tmp_path, monkeypatch):
    """AGC_AMD_GPU_ZSTD_MIN=1: even a single delta pack goes through agc_hip_zstd17_batch (here: the host build of the encoder
    the HIP kernel runs) instead of libzstd -- the archives must still be the reference's"""
    monkeypatch.setenv("AGC_AMD_GPU_ZSTD_MIN", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("name", ["toy_c1", "syn_mixed", "syn_viral", "syn_adaptive", "syn_c4_twin"])
def test_host_pipeline_with_the_references_on_the_device_entropy_stage_too(cli, name, tmp_path, monkeypatch):
    """AGC_AMD_GPU_ZSTD_REFS=1 (with AGC_AMD_GPU_ZSTD_MIN=1 and the whole share): every reference goes through agc_hip_zstd_batch as
    well -- tuple-packed at level 13, the repetitive ones as they are at level 19 (segment.h:172-255) -- instead of libzstd"""
    monkeypatch.setenv("AGC_AMD_GPU_ZSTD_REFS", "1")
    monkeypatch.setenv("AGC_AMD_GPU_ZSTD_MIN", "1")
    monkeypatch.setenv("AGC_AMD_GPU_ZSTD_SHARE", "1.0")
    monkeypatch.setenv("AGC_AMD_VERIFY_DEV_FRAMES", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "o.agc")
    r = subprocess.run([cli, "create"] + args + ["-t", "4", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert "references 13 / 19" in r.stderr and " 0 differ" in r.stderr and "differ from libzstd" not in r.stderr, r.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"]


def test_host_zstd_switch_gives_the_same_archive(cli, tmp_path, monkeypatch):
    monkeypatch.setenv("AGC_AMD_HOST_ZSTD", "1")
    args, _ = C.CONFIGS["syn_c3_twin"]
    files = C.build("syn_c3_twin", str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert hashlib.sha256(got).hexdigest() == GOLD["syn_c3_twin"]["sha256"]


@pytest.mark.parametrize("mode", ["early", "late"])
@pytest.mark.parametrize("name", ["syn_adaptive", "syn_c5_twin", "syn_mixed"])
def test_encode_overlapped_with_the_classification_gives_the_same_archive(cli, name, mode, tmp_path, monkeypatch):
    """AGC_AMD_ENCODE_OVERLAP: windows of one registration (adaptive mode; every sample handed over through the device API)
    start the encode of the key-known segments before the rest of the window is classified (agc_hip_lz_encode_begin_dev /
    _end) -- same archive; AGC_AMD_SYNC_ENTROPY=1 on top: every registration waits for its zstd parts"""
    monkeypatch.setenv("AGC_AMD_ENCODE_OVERLAP", mode)
    if mode == "late":
        monkeypatch.setenv("AGC_AMD_SYNC_ENTROPY", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("name", ["syn_mixed", "syn_shuffled", "syn_adaptive", "syn_c4_twin", "syn_viral_c"])
def test_pool_paths_of_the_host_stages_give_the_same_archive(cli, name, tmp_path, monkeypatch):
    """AGC_AMD_PAR_MIN=1: the stages that hand long lists to the worker pool at human scale (segment cut per contig, key
    lookups, missing-middle candidate search, per-group bookkeeping) take that path for these small collections too"""
    monkeypatch.setenv("AGC_AMD_PAR_MIN", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"), threads="5")
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("name,par_min", [("syn_c4_twin", "1"), ("syn_shuffled", "100000")])
def test_bookkeeping_beside_the_next_registration_gives_the_same_archive(cli, name, par_min, tmp_path, monkeypatch):
    """AGC_AMD_WINDOW_MAX=1: every file is a window of its own, so its bookkeeping (packs, in-group ids, collection records,
    archive parts) is queued to the bookkeeping thread and runs beside the next file's scan and classification -- the path
    device-resident samples (bench.py, the multi-GPU mode) always take; with the stage's own worker pool and without.
    AGC_AMD_ASYNC_BOOK=0 (the same windows, bookkeeping on the calling thread) must give the same bytes."""
    monkeypatch.setenv("AGC_AMD_WINDOW_MAX", "1")
    monkeypatch.setenv("AGC_AMD_PAR_MIN", par_min)
    monkeypatch.setenv("AGC_AMD_LAPS", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "o.agc")
    r = subprocess.run([cli, "create"] + args + ["-t", "4", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert r.stderr.count("book_and_store (queued)") == len(files), r.stderr[-2000:]
    # ... and its LZ encode is only launched by the registration; the bookkeeping task collects it (the reference sample has
    # nothing to encode)
    assert r.stderr.count("encode (in flight)") >= len(files) - 2, r.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"]
    if name == "syn_shuffled":
        monkeypatch.setenv("AGC_AMD_ASYNC_BOOK", "0")
        r = subprocess.run([cli, "create"] + args + ["-t", "4", "-o", out] + files, capture_output=True, text=True, timeout=300)
        assert "(queued)" not in r.stderr
        assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("env", [{"AGC_AMD_ASYNC_ENCODE": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_ASYNC_ENCODE": "0", "AGC_AMD_ASYNC_BOOK": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_DEV_SEGMENTS": "0", "AGC_AMD_ASYNC_ENCODE": "0", "AGC_AMD_WINDOW_MAX": "1", "AGC_AMD_SYNC_ENTROPY": "1"},
                                 {"AGC_ZSTD_LIB": "libzstd.so.1"},
                                 {"AGC_AMD_GPU_ZSTD": "0"},
                                 # round 6: the whole-sample encode collected by the registration's own task / the reference store waited for
                                 # where it is asked for, every sample on the device-launched encode and a window of its own
                                 {"AGC_AMD_EARLY_COLLECT": "0", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_PRE_LAUNCH_ENCODE": "0", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_PLACE_AHEAD": "2", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_PLACE_AHEAD": "2"},
                                 {"AGC_AMD_SPEC_FILL_AHEAD": "2", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_REF_STORE_ASYNC": "0", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_EARLY_COLLECT": "1", "AGC_AMD_REF_STORE_ASYNC": "1", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"}], ids=lambda e: "+".join(f"{k[8:] if k.startswith('AGC_AMD_') else k}={v}" for k, v in e.items()))
@pytest.mark.parametrize("name", ["syn_mixed", "syn_c4_twin"])
def test_switch_combinations_keep_the_archive(cli, name, env, tmp_path, monkeypatch):
    if name == "syn_c4_twin" and not ("AGC_AMD_DEV_SEGMENTS" in env or "AGC_AMD_ASYNC_BOOK" in env):
        pytest.skip("the 30-second twin runs with the two combinations that gate the threads; the others on syn_mixed")
    """the behaviour switches of DESIGN.md 11 that no other test sets, alone and combined with the ones that gate the threads:
    the encode collected by the calling thread, with and without the bookkeeping thread, on top of host-cut segments and a
    synchronous entropy stage; libzstd named explicitly; the device entropy stage off -- the reference's bytes every time"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("name,env", [("syn_c4_twin", {"AGC_AMD_FASTA_PACK": "0"}),  # (the configs[2] twin per contig: tests/test_gpu_archive.py)
                                      ("syn_mixed", {"AGC_AMD_FASTA_PACK_MIN": "1"}), ("syn_shuffled", {"AGC_AMD_FASTA_PACK_MIN": "1", "AGC_AMD_WINDOW_MAX": "1"}),
                                      ("syn_viral", {"AGC_AMD_FASTA_PACK_MIN": "1"})],
                         ids=["c4_twin_per_contig", "mixed_every_window", "shuffled_every_file_its_own_window", "viral_every_window"])
def test_file_path_with_and_without_the_one_pass_conversion(cli, name, env, tmp_path, monkeypatch):
    """AddSampleFiles: a window of raw contigs goes to the device as it is and is converted + packed there (agc_hip_sample_pack_fasta) --
    for every window however small (AGC_AMD_FASTA_PACK_MIN=1), or never (AGC_AMD_FASTA_PACK=0: preprocess_raw_contig per contig and a
    pack of the codes, the path before round 6); the 30 Mbp files of the twins take the one-pass conversion by default.  Same archives."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("stream", ["1", "0"])
@pytest.mark.parametrize("name", ["syn_c5_twin", "syn_snp"])
def test_entropy_batches_with_and_without_the_stream(cli, name, stream, tmp_path, monkeypatch):
    """host-only entropy batches (a few packs per window of a small collection) are streamed through the pool without a barrier
    between them (run_host_stream: later batches join the list while earlier jobs are still being compressed, parts finish in
    any order); AGC_AMD_ENTROPY_STREAM=0 is the parallel_for per batch of before.  Windows of two files: many small batches.
    The many files of these collections also go through the grouped small-file reader (eight per task)."""
    monkeypatch.setenv("AGC_AMD_ENTROPY_STREAM", stream)
    monkeypatch.setenv("AGC_AMD_WINDOW_MAX", "2")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("name", ["syn_mixed", "syn_adaptive", "syn_c5_twin"])  # (+ the configs[2] twin on the GPU: tests/test_gpu_archive.py)
def test_whole_sample_encode_from_the_device_descriptors_for_small_samples_too(cli, name, tmp_path, monkeypatch):
    """AGC_AMD_DEV_ENCODE_MIN=0 + AGC_AMD_WINDOW_MAX=1: every sample, however few segments it has, takes the path the 3 Gbp samples
    take -- its encode launched from the descriptors the device made, collected by the bookkeeping thread (by default samples of
    fewer than 2048 segments are encoded from the host's descriptors at commit time, where the library can parse in chunks)"""
    monkeypatch.setenv("AGC_AMD_DEV_ENCODE_MIN", "0")
    monkeypatch.setenv("AGC_AMD_WINDOW_MAX", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


def test_producer_tag_changes_one_stream_only(cli, tmp_path, monkeypatch):
    """AGC_AMD_PRODUCER_TAG=1: an honest producer string in file_type_info -- the archive differs from the reference's, its
    samples do not"""
    args, _ = C.CONFIGS["syn_mixed"]
    files = C.build("syn_mixed", str(tmp_path / "in"))
    plain = _create(cli, args, files, str(tmp_path / "a.agc"))
    monkeypatch.setenv("AGC_AMD_PRODUCER_TAG", "1")
    tagged = _create(cli, args, files, str(tmp_path / "b.agc"))
    assert hashlib.sha256(plain).hexdigest() == GOLD["syn_mixed"]["sha256"] and tagged != plain
    assert abs(len(tagged) - len(plain)) < 256
    monkeypatch.delenv("AGC_AMD_PRODUCER_TAG")
    names = subprocess.run([cli, "listset", str(tmp_path / "a.agc")], capture_output=True, timeout=60).stdout
    assert names and subprocess.run([cli, "listset", str(tmp_path / "b.agc")], capture_output=True, timeout=60).stdout == names
    for sn in names.decode().split()[:3]:
        a = subprocess.run([cli, "getset", str(tmp_path / "a.agc"), sn], capture_output=True, timeout=60).stdout
        # (the streamed reader too: AGC_AMD_NO_MMAP=1 reads the archive whole instead of mapping it)
        monkeypatch.setenv("AGC_AMD_NO_MMAP", "1")
        b = subprocess.run([cli, "getset", str(tmp_path / "b.agc"), sn], capture_output=True, timeout=60).stdout
        monkeypatch.delenv("AGC_AMD_NO_MMAP")
        assert a and a == b, sn


@pytest.mark.parametrize("defer_mb", [None, "0"])
@pytest.mark.parametrize("name", ["syn_mixed", "syn_adaptive"])
def test_two_ranks_in_one_process_write_the_reference_archive(name, defer_mb, tmp_path, monkeypatch):
    """the multi-GPU protocol without torch: two compressors in one process (tests/devsim/two_ranks_one_process.cpp) -- prepare
    ahead, commit in two steps (head out first, then finish + body), the writer applying records into its bookkeeping queue --
    give the reference's archive.  scripts/tsan_check.sh runs the same tool under ThreadSanitizer.
    AGC_AMD_DEFER_MAX_MB=0: the writer keeps no full pack for the distributed Close, its own entropy stage takes them as they fill."""
    if defer_mb is not None:
        monkeypatch.setenv("AGC_AMD_DEFER_MAX_MB", defer_mb)
    exe = simbuild.build_two_ranks()
    args, _ = C.CONFIGS[name]
    opt = {"-k": 31, "-l": 20, "-s": 60000, "-b": 50}
    for i in range(len(args) - 1):
        if args[i] in opt:
            opt[args[i]] = int(args[i + 1])
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "o.agc")
    r = subprocess.run([exe, out] + [str(opt[x]) for x in ("-k", "-l", "-s", "-b")] + ["1" if "-a" in args else "0"] + files,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("name", ["syn_mixed", "syn_shuffled", "syn_c4_twin", "syn_viral"])
def test_mapped_reader_gives_the_same_archive(cli, name, tmp_path, monkeypatch):
    """AGC_AMD_MAP_MIN=1: every plain input file goes through the mapped, multi-threaded reader big assemblies get"""
    monkeypatch.setenv("AGC_AMD_MAP_MIN", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    got = _create(cli, args, files, str(tmp_path / "o.agc"))
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


def test_mapped_and_stream_reader_agree_on_odd_files(cli, tmp_path, monkeypatch):
    """CR LF line ends, '>' inside a header line, a header without body at the end, no line end after the last base,
    lower case: the mapped reader must cut the same records as the stream reader"""
    import numpy as np
    rng = np.random.default_rng(3)

    def seq(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    ref = seq(40000)
    body = [ref[i:i + 70] for i in range(0, len(ref), 70)]
    d = tmp_path / "in"
    d.mkdir()
    (d / "ref.fa").write_text(">r1 desc >not a record\r\n" + "\r\n".join(body) + "\r\n>r2\n" + seq(9000) + "\n")
    mut = list(ref)
    for p in rng.integers(0, len(mut), 60):
        mut[int(p)] = "ACGT"[(("ACGT".index(mut[int(p)])) + 1) % 4]
    m = "".join(mut).lower()
    (d / "s1.fa").write_text(">r1\n" + "\n".join(m[i:i + 61] for i in range(0, len(m), 61)) + "\n>r2 x\n" + seq(5000))  # (no final newline)
    (d / "s2.fa").write_text(">r1\n" + m[:20000] + "\n>\n" + seq(100) + "\n>r9\n" + seq(3000) + "\n")  # ('>' alone: reading stops)
    files = [str(d / "ref.fa"), str(d / "s1.fa"), str(d / "s2.fa")]
    args = ["-k", "21", "-l", "17", "-s", "2000", "-b", "3"]
    a = _create(cli, args, files, str(tmp_path / "stream.agc"))
    monkeypatch.setenv("AGC_AMD_MAP_MIN", "1")
    b = _create(cli, args, files, str(tmp_path / "mapped.agc"))
    assert a == b
    ref_agc = os.path.join(ROOT, "oracle", "_ref", "agc")
    if os.path.exists(ref_agc):  # (this container: the reference CLI reads the same records)
        subprocess.run([ref_agc, "create"] + args + ["-t", "2", "-o", str(tmp_path / "ref.agc")] + files, check=True, capture_output=True, timeout=120)
        assert open(tmp_path / "ref.agc", "rb").read() == a


def test_host_pipeline_is_thread_independent(cli, tmp_path):
    args, _ = C.CONFIGS["syn_adaptive"]
    files = C.build("syn_adaptive", str(tmp_path / "in"))
    a = _create(cli, args, files, str(tmp_path / "a.agc"), threads="1")
    b = _create(cli, args, files, str(tmp_path / "b.agc"), threads="7")
    assert a == b


def test_product_library_is_not_the_simulator():
    """the product build must not pick the stand-in up: libagc_hip.so in agc_amd/ comes from hipcc and carries gfx950 code"""
    from agc_amd import build
    lib = build.build()
    assert os.path.dirname(lib) == os.path.join(ROOT, "agc_amd")
    blob = open(lib, "rb").read()
    assert b"gfx950" in blob and b"devsim" not in blob


_PACKED_CHILD = r"""
import ctypes as ct, json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from agc_amd import capi, fasta, host
from tests.devsim import build as simbuild
opt, files, out = json.loads(sys.argv[2]), json.loads(sys.argv[3]), sys.argv[4]
lib = host.bind(ct.CDLL(simbuild.SIM_HOST))
sim = ct.CDLL(simbuild.SIM_HIP)
sim.agc_hip_packed_words_bytes.restype = ct.c_uint64
sim.agc_hip_packed_words_bytes.argtypes = [ct.c_uint64]
sim.agc_hip_packed_index_bytes.restype = ct.c_uint64
sim.agc_hip_packed_index_bytes.argtypes = [ct.c_uint64]
sim.agc_hip_pack_dev.argtypes = [ct.c_void_p] * 2 + [ct.c_uint64] + [ct.c_void_p] * 3 + [ct.c_uint64, ct.c_void_p]
cmp_ = host.Compressor(lib=lib)
cmp_.create(out, pack_cardinality=opt["-b"], k=opt["-k"], ref_file=files[0], segment_size=opt["-s"], min_match_len=opt["-l"], n_threads=2)
ctx = ct.c_void_p(1)  # the stand-in's pack entry point only checks for a context pointer
packed, keep = [], []
for f in files:
    names, codes, off = fasta.read_codes(f)
    n = codes.size
    words = np.zeros(int(sim.agc_hip_packed_words_bytes(n)) // 4, np.uint32)
    index = np.zeros(int(sim.agc_hip_packed_index_bytes(n)) // 4, np.int32)
    esc = np.zeros((n // 1024 + 2) * 1024, np.uint8)
    cnt = np.zeros(1, np.uint64)
    assert sim.agc_hip_pack_dev(ctx, codes.ctypes.data, n, words.ctypes.data, index.ctypes.data, esc.ctypes.data, n // 1024 + 2, cnt.ctypes.data) == 0
    keep.append((words, index, esc))
    packed.append((fasta.sample_name(f), names, capi.Packed(words.ctypes.data, index.ctypes.data, esc.ctypes.data, n), off))
announce = int(sys.argv[5])  # 0: never; 1: the next sample before every add; 2: sometimes the WRONG one (must be dropped, not used)
for i, (sn, names, pk, off) in enumerate(packed):
    if announce and i + 1 < len(packed) and all(len(x[1]) for x in packed):
        j = i + 1 if (announce == 1 or i % 2 == 0 or i + 2 >= len(packed)) else i + 2
        cmp_.set_next_sample_packed_dev(packed[j][2], packed[j][3])
    cmp_.add_sample_packed_dev(sn, names, pk, off)
cmp_.close()
cmp_.close_handle()
"""


@pytest.mark.parametrize("overlap,announce", [("off", 0), ("early", 0), ("off", 1), ("off", 2)])
@pytest.mark.parametrize("name", ["syn_mixed", "syn_snp", "syn_shuffled"])
def test_samples_in_the_packed_layout_give_the_reference_archive(cli, name, overlap, announce, tmp_path, monkeypatch):
    """AddSamplePackedDevice: every sample handed over in the 2-bit layout (escaped blocks for N runs / IUPAC codes, contigs at
    arbitrary symbol offsets); the archive must be the one the reference CLI writes from the FASTA files.  (In a child process:
    the host library linked with the CPU stand-in must not meet the product's libagc_hip.so of the same name.)"""
    import json
    import sys
    args, _ = C.CONFIGS[name]
    opt = {"-k": 31, "-l": 20, "-s": 60000, "-b": 50}
    for i in range(len(args) - 1):
        if args[i] in opt:
            opt[args[i]] = int(args[i + 1])
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "packed.agc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.setenv("AGC_AMD_ENCODE_OVERLAP", overlap)  # (one registration per window here: the overlapped encode runs)
    # announce: SetNextSamplePackedDevice before every add (the next sample's expansion + scan queued ahead), also with a wrong guess
    subprocess.check_call([sys.executable, "-c", _PACKED_CHILD, root, json.dumps(opt), json.dumps(files), out, str(announce)])
    got = open(out, "rb").read()
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("name", ["syn_mixed", "syn_c4_twin"])
def test_segments_from_the_device_or_cut_on_the_host_give_the_same_archive(cli, name, tmp_path, monkeypatch):
    """Single-registration windows take their segments, keys and group look-ups from the device (agc_hip_segments_packed; the
    encode of the segments whose group is known is launched there too) -- the laps say so; AGC_AMD_DEV_SEGMENTS=0 (hits to the
    host, cut and look-up there) must write the same bytes."""
    monkeypatch.setenv("AGC_AMD_WINDOW_MAX", "1")
    monkeypatch.setenv("AGC_AMD_LAPS", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "o.agc")
    r = subprocess.run([cli, "create"] + args + ["-t", "4", "-o", out] + files, capture_output=True, text=True, timeout=300)
    if "-a" not in args and "-c" not in args:
        assert r.stderr.count("scan + segments (device)") == len(files), r.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"]
    monkeypatch.setenv("AGC_AMD_DEV_SEGMENTS", "0")
    r = subprocess.run([cli, "create"] + args + ["-t", "4", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert "segments (device)" not in r.stderr
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"]
