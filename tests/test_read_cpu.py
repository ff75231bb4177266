"""Read side on CPU: libagc_read.so / `agc_amd getset|getctg|list*` against
 - the committed reference-built toy archive (tests/golden/toy_c1_reference.agc) and its input FASTA files,
 - archives the reference CLI (oracle/_ref/agc, when prebuilt) writes for every collection of tests/collections.py:
   decoded samples must equal the input sequences (round trip) and the reference's own `getset` / `getctg` text.
The archives `agc_amd create` writes are byte-identical to these (tests/test_gpu_archive.py), so the same reader
covers both sides."""
import ctypes as C
import os
import subprocess

import pytest

from tests import collections as COLL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AGC_AMD = os.path.join(ROOT, "agc_amd", "bin", "agc_amd")
REF_AGC = os.path.join(ROOT, "oracle", "_ref", "agc")
REF_ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
TOY_AGC = os.path.join(ROOT, "tests", "golden", "toy_c1_reference.agc")
TOY = os.path.join(ROOT, "tests", "golden", "toy_ex")


@pytest.fixture(scope="module")
def rd():
    from agc_amd import build, reader
    build.build_read()
    return reader


def parse_fasta(path):
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    out, name, seq = [], None, []
    with op(path, "rt") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(seq).upper()))
                name, seq = line[1:], []
            else:
                seq.append(line)
    if name is not None:
        out.append((name, "".join(seq).upper()))
    return out


def fasta_text(contigs, line=80):
    t = []
    for n, s in contigs:
        t.append(">" + n + "\n")
        for i in range(0, len(s), line):
            t.append(s[i:i + line] + "\n")
    return "".join(t).encode()


def test_library_exports_every_declared_symbol(rd):
    import re
    hdr = open(os.path.join(ROOT, "include", "agc_read.h")).read()
    declared = sorted(set(re.findall(r"\b(agc_[a-z_]+)\s*\(", hdr)) - {"agc_t"})
    assert declared == sorted(rd.SYMBOLS)
    L = C.CDLL(rd.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s


def test_toy_archive_api(rd):
    a = rd.CAGCFile()
    assert not a.Close()
    assert a.Open(TOY_AGC, True)
    assert not a.Open(TOY_AGC, True)
    assert a.NSample() == 4
    assert a.GetReferenceSample() == "ref"
    v = rd.StringVector()
    assert a.ListSample(v) == 0 and list(v) == ["a", "b", "c", "ref"]
    assert a.GetParams() == {"k": 25, "min_match_len": 17, "pack_cardinality": 50, "segment_size": 60000}
    for s in v:
        want = parse_fasta(os.path.join(TOY, s + ".fa"))
        assert a.NCtg(s) == len(want)
        assert a.ListCtg(s) == [n for n, _ in want]
        for n, seq in want:
            assert a.GetCtgLen(s, n) == len(seq)
            assert a.GetCtgSeq(s, n, -1, -1) == seq
            short = n.split()[0]
            assert a.GetCtgSeq(s, short, 0, len(seq) - 1) == seq
            if len(seq) > 14:
                assert a.GetCtgSeq(s, n, 8, 12) == seq[8:13]
                assert a.GetCtgSeq(s, n, len(seq) - 3, len(seq) + 50) == seq[-3:]
        assert a.GetSampleFasta(s) == fasta_text(want)
    assert a.NCtg("nope") == -1 and a.GetCtgLen("nope", "x") < 0 and a.GetCtgLen("ref", "no-such-contig") == -1
    assert a.Close() and a.NSample() == -1


def test_unknown_file_and_garbage_are_rejected(rd, tmp_path):
    a = rd.CAGCFile()
    assert not a.Open(str(tmp_path / "missing.agc"))
    p = tmp_path / "junk.agc"
    p.write_bytes(b"\x00" * 100)
    assert not a.Open(str(p))
    p.write_bytes(open(TOY_AGC, "rb").read()[:-9])
    assert not a.Open(str(p))


def test_contig_name_without_sample_must_be_unique(rd):
    a = rd.CAGCFile()
    assert a.Open(TOY_AGC)
    names = {}
    for s in a.ListSample():
        for n in a.ListCtg(s):
            names.setdefault(n.split()[0], []).append(s)
    for n, ss in names.items():
        if len(ss) == 1:
            assert a.GetCtgLen("", n) == a.GetCtgLen(ss[0], n) > 0
            assert a.GetCtgSeq(n, -1, -1) == a.GetCtgSeq(ss[0], n, -1, -1)
        else:
            assert a.GetCtgLen("", n) == -2  # lib-cxx.cpp / agc_decompressor_lib.cpp:113-121
    a.Close()


def _cli(args, **kw):
    from agc_amd import build
    build.build_host()
    return subprocess.run([AGC_AMD] + args, capture_output=True, timeout=300, **kw)


def test_cli_list_and_get_on_toy():
    assert _cli(["listset", TOY_AGC]).stdout == b"a\nb\nc\nref\n"
    assert _cli(["listref", TOY_AGC]).stdout == b"ref"
    want = parse_fasta(os.path.join(TOY, "a.fa"))
    assert _cli(["getset", TOY_AGC, "a"]).stdout == fasta_text(want)
    assert _cli(["getset", "-l", "40", TOY_AGC, "a"]).stdout == fasta_text(want, 40)
    lc = _cli(["listctg", TOY_AGC, "a", "ref"]).stdout.decode().split("\n")
    assert lc[0] == "a" and lc[1] == "   " + want[0][0]
    n, s = want[0]
    short = n.split()[0]
    assert _cli(["getctg", TOY_AGC, f"{short}@a"]).stdout == fasta_text([(n, s)])
    assert _cli(["getctg", TOY_AGC, f"{short}@a:3-10"]).stdout == fasta_text([(n + ":3-10", s[3:11])])
    r = _cli(["getctg", TOY_AGC, "nothing@a"])
    assert r.returncode == 0 and r.stdout == b"" and b"no sample:contig" in r.stderr


def test_cli_info_prints_what_the_reference_prints():
    got = _cli(["info", "-v", "1", TOY_AGC]).stderr.decode()
    assert got.startswith("No. samples      : 4\nk-mer length     : 25\nMin. match length: 17\nSegment size     : 60000\n"
                          "Batch size       : 50\nReference name   : ref\nCommand lines:\nFile type info:\n")
    assert "  file_version_major : 3\n" in got and "  producer : agc\n" in got
    if os.path.exists(REF_AGC):
        want = subprocess.run([REF_AGC, "info", "-v", "1", TOY_AGC], capture_output=True, env=REF_ENV).stderr.decode()
        assert want.startswith(got)  # the reference adds its "Completed in" footer


@pytest.mark.parametrize("name", list(COLL.CONFIGS))
def test_reference_archives_round_trip_and_match_reference_getset(rd, name, tmp_path):
    if not os.path.exists(REF_AGC):
        pytest.skip("oracle/_ref/agc not prebuilt")
    args, _ = COLL.CONFIGS[name]
    files = COLL.build(name, str(tmp_path / "in"))
    arc = str(tmp_path / "ref.agc")
    subprocess.run([REF_AGC, "create"] + args + ["-t", "4", "-o", arc] + files, check=True, capture_output=True, timeout=300, env=REF_ENV)
    a = rd.CAGCFile()
    assert a.Open(arc)
    ref_list = subprocess.run([REF_AGC, "listset", arc], check=True, capture_output=True, env=REF_ENV).stdout.decode().split()
    assert a.ListSample() == ref_list
    ref_name = subprocess.run([REF_AGC, "listref", arc], check=True, capture_output=True, env=REF_ENV).stdout.decode()
    assert a.GetReferenceSample() == ref_name
    # what went in: in -c mode every contig is its own sample (named by the contig's short name)
    if "-c" in args:
        inputs = {}
        for f in files:
            for n, s in parse_fasta(f):
                inputs.setdefault(n.split()[0], []).append((n, s))
    else:
        inputs = {}
        for f in files:
            sn = os.path.basename(f)
            for suf in (".gz", ".fa", ".fasta", ".fna"):
                sn = sn[:-len(suf)] if sn.endswith(suf) else sn
            inputs[sn] = parse_fasta(f)
    assert sorted(inputs) == sorted(ref_list)
    for i, sn in enumerate(ref_list):
        want = inputs[sn]
        got = a.GetSampleFasta(sn)
        assert got == fasta_text(want), sn
        assert a.ListCtg(sn) == [n for n, _ in want]
        if i % 7 == 0:  # the reference's own decoder on a subset (it is slower to spawn than to decode)
            ref_txt = subprocess.run([REF_AGC, "getset", arc, sn], check=True, capture_output=True, env=REF_ENV).stdout
            assert got == ref_txt, sn
            assert _cli(["getset", arc, sn]).stdout == ref_txt
        n, s = want[-1]
        if len(s) > 200:
            q = f"{n.split()[0]}@{sn}:{len(s) // 3}-{len(s) // 3 + 150}"
            assert a.GetCtgSeq(sn, n, len(s) // 3, len(s) // 3 + 150) == s[len(s) // 3: len(s) // 3 + 151]
            if i % 7 == 0:
                ref_txt = subprocess.run([REF_AGC, "getctg", arc, q], check=True, capture_output=True, env=REF_ENV).stdout
                assert _cli(["getctg", arc, q]).stdout == ref_txt
    a.Close()
