"""GPU end-to-end parity: `agc_amd create` (HIP kernels + host ordering contract + libzstd) must write
byte-identical archives to the reference CLI on the same inputs -- against the sha256 recorded from
the reference in tests/golden/archives.json, and against oracle/_ref/agc live when it was prebuilt."""
import hashlib
import json
import os
import subprocess

import pytest

from tests import collections as C

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AGC_AMD = os.path.join(ROOT, "agc_amd", "bin", "agc_amd")
REF_AGC = os.path.join(ROOT, "oracle", "_ref", "agc")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "archives.json")))


@pytest.mark.parametrize("name", list(C.CONFIGS))
def test_archive_bit_identical(name, tmp_path):
    from agc_amd import build
    from tests import agc_container
    build.build_host()
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "amd.agc")
    r = subprocess.run([AGC_AMD, "create"] + args + ["-t", "8", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr[-2000:]
    got = open(out, "rb").read()
    if os.path.exists(REF_AGC):
        ref = str(tmp_path / "ref.agc")
        subprocess.run([REF_AGC, "create"] + args + ["-t", "4", "-o", ref] + files, check=True, capture_output=True, timeout=300)
        want = open(ref, "rb").read()
        assert hashlib.sha256(want).hexdigest() == GOLD[name]["sha256"], "reference build no longer matches its recorded golden"
        if got != want:
            pytest.fail("archive differs from the reference's:\n" + "\n".join(agc_container.diff(want, got)) + "\nstderr: " + r.stderr[-1500:])
    assert len(got) == GOLD[name]["size"], r.stderr[-1500:]
    assert hashlib.sha256(got).hexdigest() == GOLD[name]["sha256"]


@pytest.mark.parametrize("mode", ["early", "late"])
@pytest.mark.parametrize("name", ["syn_adaptive", "syn_c5_twin", "syn_c3_twin"])
def test_archive_with_the_encode_on_the_second_stream(name, mode, tmp_path, monkeypatch):
    """AGC_AMD_ENCODE_OVERLAP=early|late: windows of one registration encode their key-known segments on the context's second
    HIP stream (agc_hip_lz_encode_begin_dev / _end) while estimates, cost vectors and index builds run on the first one;
    with AGC_AMD_SYNC_ENTROPY=1 on top every registration also waits for its zstd parts.  Same bytes as the reference."""
    from agc_amd import build
    build.build_host()
    monkeypatch.setenv("AGC_AMD_ENCODE_OVERLAP", mode)
    if mode == "late":
        monkeypatch.setenv("AGC_AMD_SYNC_ENTROPY", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "amd.agc")
    r = subprocess.run([AGC_AMD, "create"] + args + ["-t", "8", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"], r.stderr[-1500:]


@pytest.mark.parametrize("name", ["syn_mixed", "syn_c3_twin", "syn_c4_twin"])
def test_archive_with_every_reference_on_the_device_entropy_stage(name, tmp_path, monkeypatch):
    """AGC_AMD_GPU_ZSTD_REFS=1: every call that brings a reference hands it to the device encoder (levels 13 on the tuple-packed
    symbols / 19 for repetitive ones, agc_hip_zstd_batch) -- the default only does so for calls with 512 references and more, which
    these small collections never bring -- and AGC_AMD_GPU_ZSTD_MIN=1 sends every delta pack there too: the archive must still be
    the reference CLI's."""
    from agc_amd import build
    build.build_host()
    monkeypatch.setenv("AGC_AMD_GPU_ZSTD_REFS", "1")
    monkeypatch.setenv("AGC_AMD_GPU_ZSTD_MIN", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "amd.agc")
    r = subprocess.run([AGC_AMD, "create"] + args + ["-t", "8", "-v", "1", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr[-2000:]
    assert "entropy stage: device" in r.stderr, r.stderr[-1500:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"], r.stderr[-1500:]


GOLD_APPEND = json.load(open(os.path.join(ROOT, "tests", "golden", "archives_append.json")))


@pytest.mark.parametrize("plan", list(C.APPEND_PLANS))
def test_append_bit_identical(plan, tmp_path):
    """`agc_amd create` + `agc_amd append` sequences on the GPU: every archive equals the reference CLI's (recorded sha256)"""
    from agc_amd import build
    build.build_host()
    got = C.run_append_plan(AGC_AMD, plan, str(tmp_path), threads="8")
    want = GOLD_APPEND[plan]
    assert [len(x) for x in got] == [w["size"] for w in want]
    assert [hashlib.sha256(x).hexdigest() for x in got] == [w["sha256"] for w in want]


def test_cli_without_reference_file_reports_and_exits_zero(tmp_path):
    r = subprocess.run([AGC_AMD, "create", "-o", str(tmp_path / "x.agc"), str(tmp_path / "missing.fa")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Cannot" in r.stderr


def _run_amd(args, files, out, threads="8"):
    r = subprocess.run([AGC_AMD, "create"] + args + ["-t", threads, "-o", out] + files, capture_output=True, text=True, timeout=600)
    assert os.path.exists(out), r.stderr[-2000:]
    return open(out, "rb").read()


def test_survey_scale_collection_vs_reference(tmp_path):
    """SURVEY App. B.3 shape: 20 Mbp reference in 4 contigs + 6 samples at d = 1e-3, default parameters
    (s = 60000): ~2400 placed segments, missing-middle splits, reference streams ~5 MB.  Needs the
    reference binary (prebuilt oracle/_ref); byte-for-byte."""
    if not os.path.exists(REF_AGC):
        pytest.skip("oracle/_ref/agc not prebuilt")
    import numpy as np
    from agc_amd import build, synth
    from tests import agc_container
    build.build_host()
    rng = np.random.default_rng(12345)
    ref = [synth.random_seq(rng, 5_000_000) for _ in range(4)]
    names = [f"chr{i + 1}" for i in range(4)]
    d = tmp_path / "in"
    d.mkdir()
    files = [str(d / "ref.fa")]
    synth.to_fasta(files[0], ref, names)
    for s in range(6):
        fn = str(d / f"s{s}.fa")
        synth.to_fasta(fn, [synth.mutate(rng, c, 1e-3) for c in ref], names)
        files.append(fn)
    want_fn = str(tmp_path / "ref.agc")
    subprocess.run([REF_AGC, "create", "-t", "8", "-o", want_fn] + files, check=True, capture_output=True, timeout=600)
    want = open(want_fn, "rb").read()
    got = _run_amd([], files, str(tmp_path / "amd.agc"))
    if got != want:
        pytest.fail("\n".join(agc_container.diff(want, got)))


def test_archive_is_deterministic_and_thread_independent(tmp_path):
    from agc_amd import build
    build.build_host()
    args, _ = C.CONFIGS["syn_mixed"]
    files = C.build("syn_mixed", str(tmp_path / "in"))
    a = _run_amd(args, files, str(tmp_path / "a.agc"), threads="1")
    b = _run_amd(args, files, str(tmp_path / "b.agc"), threads="16")
    assert a == b and hashlib.sha256(a).hexdigest() == GOLD["syn_mixed"]["sha256"]


def test_gz_input_and_file_list(tmp_path):
    """.gz inputs (src/core/genome_io.cpp via gz_wrapper) and -i <list> give the same archive"""
    import gzip
    import shutil
    from agc_amd import build
    build.build_host()
    args, _ = C.CONFIGS["syn_shuffled"]
    files = C.build("syn_shuffled", str(tmp_path / "in"))
    gz = []
    for f in files:
        g = f + ".gz"
        with open(f, "rb") as fi, gzip.open(g, "wb", compresslevel=1) as fo:
            shutil.copyfileobj(fi, fo)
        gz.append(g)
    a = _run_amd(args, gz, str(tmp_path / "gz.agc"))
    # sample names are the file stems minus .fa/.gz suffixes, so the archive is identical to the plain one
    assert hashlib.sha256(a).hexdigest() == GOLD["syn_shuffled"]["sha256"]
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(files[1:]) + "\n")
    out = str(tmp_path / "lst.agc")
    r = subprocess.run([AGC_AMD, "create"] + args + ["-i", str(lst), "-o", out, files[0]], capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD["syn_shuffled"]["sha256"]


@pytest.mark.parametrize("concat", [False, True])
def test_baseline_config1_full_size(tmp_path, concat):
    """BASELINE.json configs[1] at its full size: 1000 genomes x 30 kb, 1 % SNPs from one reference, default
    parameters -- one file per genome, and the same collection as one concatenated file (-c).  Exercises the
    speculation window (hundreds of registrations classified per GPU pass).  Byte-for-byte against the
    reference CLI (needs the prebuilt oracle/_ref/agc)."""
    if not os.path.exists(REF_AGC):
        pytest.skip("oracle/_ref/agc not prebuilt")
    import numpy as np
    from agc_amd import build, synth
    from tests import agc_container
    build.build_host()
    rng = np.random.default_rng(2)
    ref = synth.random_seq(rng, 30_000)
    genomes = [ref] + [synth.mutate(rng, ref, 0.01) for _ in range(999)]
    d = tmp_path / "in"
    d.mkdir()
    if concat:
        files = [str(d / "ref.fa"), str(d / "all.fa")]
        synth.to_fasta(files[0], [genomes[0]], ["MN000000.1 synthetic genome 0"])
        synth.to_fasta(files[1], genomes[1:], [f"MN{i:06d}.1 synthetic genome {i}" for i in range(1, 1000)])
    else:
        files = []
        for i, g in enumerate(genomes):
            fn = str(d / f"g{i:04d}.fa")
            synth.to_fasta(fn, [g], [f"MN{i:06d}.1 synthetic genome {i}"])
            files.append(fn)
    args = ["-c"] if concat else []
    want_fn = str(tmp_path / "ref.agc")
    subprocess.run([REF_AGC, "create", "-t", "8"] + args + ["-o", want_fn] + files, check=True, capture_output=True, timeout=600)
    want = open(want_fn, "rb").read()
    got = _run_amd(args, files, str(tmp_path / "amd.agc"))
    if got != want:
        pytest.fail("\n".join(agc_container.diff(want, got)))


@pytest.mark.parametrize("name", ["syn_mixed", "syn_adaptive", "syn_c5_twin", "syn_c3_twin", "syn_viral"])
def test_whole_sample_encode_from_the_device_descriptors_for_small_samples_too(name, tmp_path, monkeypatch):
    """AGC_AMD_DEV_ENCODE_MIN=0 + AGC_AMD_WINDOW_MAX=1: every sample, however small, has its encode launched from the descriptors
    the device made (agc_hip_segments_encode_known) and collected by the bookkeeping thread -- the path of the 3 Gbp samples, on
    the real kernels; by default small samples are encoded from the host's descriptors at commit time"""
    from agc_amd import build
    build.build_host()
    monkeypatch.setenv("AGC_AMD_DEV_ENCODE_MIN", "0")
    monkeypatch.setenv("AGC_AMD_WINDOW_MAX", "1")
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "amd.agc")
    r = subprocess.run([AGC_AMD, "create"] + args + ["-t", "8", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"], r.stderr[-1500:]


# sha256 / size of the archive the reference CLI (oracle/_ref/agc, libzstd 1.4.9) wrote for BASELINE configs[2] at FULL size --
# the seeded 3 Gbp GRCh38-shaped reference + ONE 3 Gbp sample at d = 1e-3, -k 31 -l 15 -b 100 -- recorded by
@pytest.mark.parametrize("env", [{"AGC_AMD_FASTA_PACK": "0"}, {"AGC_AMD_FASTA_PACK_MIN": "1", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_EARLY_COLLECT": "0", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_REF_STORE_ASYNC": "0", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_HIP_UPLOAD_RING_MB": "1", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_PRE_LAUNCH_ENCODE": "0", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"},
                                 {"AGC_AMD_PLACE_AHEAD": "2", "AGC_AMD_DEV_ENCODE_MIN": "0"},
                                 {"AGC_AMD_SPEC_FILL_AHEAD": "2", "AGC_AMD_DEV_ENCODE_MIN": "0", "AGC_AMD_WINDOW_MAX": "1"}],
                         ids=["per_contig_conversion", "one_pass_conversion_of_every_file", "encode_collected_by_the_registrations_task", "reference_store_waited_for",
                              "upload_ring_of_1_MB_wraps_many_times", "encode_launched_behind_the_segment_table",
                              "placement_beside_the_wait_for_the_split_points", "table_of_speculative_deltas_filled_by_a_helper"])
@pytest.mark.parametrize("name", ["syn_c3_twin", "syn_mixed", "syn_adaptive"])
def test_round6_switches_keep_the_archive(name, env, tmp_path, monkeypatch):
    """the file path with and without the one-pass FASTA conversion (agc_hip_sample_pack_fasta), the whole-sample encode collected by an
    early task or by the registration's own, the reference store on a stream of its own or waited for, a pinned upload ring so small that
    its head comes back to every part (and waits for the part's events) many times: the real kernels, every sample on
    the device-launched encode and a window of its own where that matters -- the reference's archive every time (syn_adaptive is the
    collection that caught the packed buffers being packed again under the reference-store stream)"""
    from agc_amd import build
    build.build_host()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    args, _ = C.CONFIGS[name]
    files = C.build(name, str(tmp_path / "in"))
    out = str(tmp_path / "amd.agc")
    r = subprocess.run([AGC_AMD, "create"] + args + ["-t", "8", "-o", out] + files, capture_output=True, text=True, timeout=300)
    assert os.path.exists(out), r.stderr[-2000:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == GOLD[name]["sha256"], r.stderr[-1500:]


# scripts/c3_full_identity.py on the GPU box (profiles/r4/c3_full_size_identity_1_sample_against_reference_cli_run.log: 256 s of
# reference-CLI time, which is why the test compares with the recorded hash instead of running the CLI again)
C3_FULL_1_SAMPLE = ("46b81e041ac68005a743d229bca09bf1d809d161fc3a145f6f97cba7858e6d9f", 767396631)


def test_configs2_at_full_size_equals_the_reference_archive():
    """BASELINE configs[2] at full contig size through the product path bench.py times (samples resident in the 2-bit layout,
    agc_cmp_add_sample_packed_dev, device segments, two encode lanes, GPU entropy stage): the archive has the reference CLI's
    recorded sha256, and every frame the device entropy stage returned equals libzstd's (AGC_AMD_VERIFY_DEV_FRAMES=1)."""
    import re
    import sys
    env = dict(os.environ, AGC_AMD_VERIFY_DEV_FRAMES="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "c3_full_identity.py"), "3.0", "1", C3_FULL_1_SAMPLE[0]],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    assert f"agc_amd {C3_FULL_1_SAMPLE[1]} {C3_FULL_1_SAMPLE[0]}" in r.stdout
    checks = re.findall(r"verify: (\d+) device frames .*?: (\d+) differ", r.stderr)
    assert checks and sum(int(n) for n, _ in checks) > 10000, r.stderr[-1500:]  # (the reference sample's 50 k references go through levels 13 / 19 there)
    assert all(int(bad) == 0 for _, bad in checks), checks
