"""Whole-archive parity on random small collections (tests/fuzz.py): parameters, contig edits (substitutions, indels, N runs, IUPAC,
reverse complements, fragments), rearranged / missing / novel / duplicated contigs, contigs shorter than k, -c, -a and create/append
splits are all drawn from the seed.  Every archive must equal the reference CLI's byte for byte; the reference is run live
(oracle/_ref/agc, prebuilt), so the tests skip where it is absent.  Where the reference itself dies (it segfaults on some
`append -c` inputs) the comparison stops at that step and agc_amd only has to survive.
CPU: host pipeline on the device stand-in; GPU: the product CLI."""
import os

import pytest

from tests import fuzz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_AGC = os.path.join(ROOT, "oracle", "_ref", "agc")
REF_ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))

# seeds that once exposed a difference stay in the list: 31..84 = two contigs with the same name inside one sample (collection
# records), 21201 = the same with -b 1 (the second copy's items for NEW groups never enter the reference's std::set)
CPU_SEEDS = [0, 1, 3, 5, 7, 11, 13, 31, 35, 38, 57, 70, 78, 84, 887, 1091, 21201] + list(range(100, 112))
GPU_SEEDS = [0, 1, 2, 3, 5, 6, 9, 13, 14, 17, 31, 35, 38, 57, 21201] + list(range(200, 216))


def _check(cli, seed, tmp_path):
    case = fuzz.make_case(seed, str(tmp_path / "in"))
    # one thread: with two equally named contigs in a sample the reference's collection records depend on which worker
    # thread writes last (store_segments, agc_compressor.cpp:989-1050); single-threaded it is the lowest group id
    want, _ = fuzz.run_case(REF_AGC, case, str(tmp_path), "ref", threads="1", env=REF_ENV)
    got, errs = fuzz.run_case(cli, case, str(tmp_path), "amd")
    assert None not in got, "agc_amd died: " + errs[-1][-1000:]
    n = len(want) - 1 if want[-1] is None else len(want)
    assert n >= 1 and all(want[:n]), "the reference produced nothing for this case"
    for i in range(n):
        if got[i] != want[i]:
            used = case["files"][:sum(case["steps"][:i + 1])]
            ref_ok = fuzz.archive_round_trips(str(tmp_path / f"ref_{i}.agc"), used)
            amd_ok = fuzz.archive_round_trips(str(tmp_path / f"amd_{i}.agc"), used)
            # (the reference loses the appended contigs' records in `append -c` onto a completely filled batch)
            assert not ref_ok and amd_ok, f"seed {seed} step {i}: {' '.join(case['args'] + case['carry'])} steps {case['steps']}"
            break


@pytest.mark.parametrize("seed", [0, 1, 3, 5, 13, 100, 101, 104, 107, 110])
def test_fuzz_read_side(seed, tmp_path):
    """random getset / getctg / listctg queries on a reference-built archive: agc_amd's read side prints what the reference prints"""
    if not os.path.exists(REF_AGC):
        pytest.skip("oracle/_ref/agc not prebuilt")
    from agc_amd import build
    build.build_host()
    case = fuzz.make_case(seed, str(tmp_path / "in"))
    want, _ = fuzz.run_case(REF_AGC, case, str(tmp_path), "ref", threads="1", env=REF_ENV)
    last = max(i for i, x in enumerate(want) if x)
    bad = fuzz.read_side_matches(REF_AGC, build.HOST_BIN, str(tmp_path / f"ref_{last}.agc"), seed, env=REF_ENV)
    assert bad is None, bad


@pytest.mark.parametrize("seed", CPU_SEEDS)
def test_fuzz_host_pipeline(seed, tmp_path):
    if not os.path.exists(REF_AGC):
        pytest.skip("oracle/_ref/agc not prebuilt")
    from tests.devsim import build as simbuild
    _check(simbuild.build(), seed, tmp_path)


@pytest.mark.parametrize("ranks,first", [(2, 100), (3, 300)])
def test_fuzz_n_ranks_in_one_process(ranks, first):
    """the multi-GPU single-archive protocol (tests/devsim/two_ranks_one_process.cpp: W compressors in one process, prefetching
    schedule, two-step commit, the writer's bookkeeping queue, adaptive mode included) on random collections against the live
    reference CLI -- scripts/fuzz_two_ranks.py runs the long campaigns"""
    if not os.path.exists(REF_AGC):
        pytest.skip("oracle/_ref/agc not prebuilt")
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_two_ranks.py"), "--from", str(first), "--count", "30", "--ranks", str(ranks)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
    compared = int(r.stdout.rsplit("compared: ", 1)[1].split(",")[0])
    assert compared >= 15, r.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", GPU_SEEDS)
def test_fuzz_gpu(seed, tmp_path):
    if not os.path.exists(REF_AGC):
        pytest.skip("oracle/_ref/agc not prebuilt")
    from agc_amd import build
    build.build_host()
    _check(build.HOST_BIN, seed, tmp_path)
