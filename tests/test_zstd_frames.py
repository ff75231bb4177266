"""CPU parity of the zstd level-17 encoder (agc_amd/csrc/zstd/*.h, compiled for the host by tests/zstd_host) with the image's
libzstd 1.4.9: the parser's sequences against ZSTD_generateSequences, whole frames against ZSTD_compressCCtx(level 17), the
restated parameter selection against ZSTD_getCParams.  The GPU build of the same headers is checked by tests/test_gpu_zstd.py."""
import ctypes as C

import numpy as np
import pytest

from tests import zstd_cases as ZC
from tests.zstd_host import build as zbuild


@pytest.fixture(scope="module")
def zs():
    if ZC.libzstd().ZSTD_versionNumber() != 10409:
        pytest.skip("parity is pinned against libzstd 1.4.9")
    H = C.CDLL(zbuild.build())
    H.zs_host_parse.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    H.zs_host_compress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    H.zs_host_compress.restype = C.c_uint32
    H.zs_host_compress2.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    H.zs_host_compress2.restype = C.c_uint32
    H.zs_host_compress_grp.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    H.zs_host_compress_grp.restype = C.c_uint32
    return H


def _my_frame(H, data, loop_nest=0):
    n = len(data)
    cp = np.array(ZC.ref_cparams(n), np.uint32)
    out = np.zeros(n + 64, np.uint8)
    k = H.zs_host_compress2(bytes(data), n, cp.ctypes.data, out.ctypes.data, loop_nest)
    return out[:k].tobytes()


def _my_frame_grp(H, data, g):
    n = len(data)
    cp = np.array(ZC.ref_cparams(n), np.uint32)
    out = np.zeros(n + 64, np.uint8)
    k = H.zs_host_compress_grp(bytes(data), n, cp.ctypes.data, out.ctypes.data, g)
    return out[:k].tobytes()


def _my_sequences(H, data):
    """the parser's (offCode, litLength, matchLength) resolved to libzstd's (offset, litLength, matchLength, rep) form"""
    n = len(data)
    cp = np.array(ZC.ref_cparams(n), np.uint32)
    out = np.zeros(3 * (n // 3 + 16), np.uint32)
    ll = np.zeros(1, np.uint32)
    k = H.zs_host_parse(bytes(data), n, cp.ctypes.data, out.ctypes.data, n // 3 + 16, ll.ctypes.data)
    rep, res = [1, 4, 8], []
    for off_code, l, m in out[:3 * k].reshape(-1, 3).tolist():
        if off_code >= 3:
            raw, r = off_code - 2, 0
            rep = [raw, rep[0], rep[1]]
        else:
            r = off_code + 1
            rc = off_code + (1 if l == 0 else 0)
            if rc == 0:
                raw = rep[0]
            else:
                raw = rep[0] - 1 if rc == 3 else rep[rc]
                rep = [raw, rep[0], rep[1]] if rc >= 2 else [raw, rep[0], rep[2]]
        res.append((raw, l, m, r))
    res.append((0, int(ll[0]), 0, 0))
    return res


@pytest.mark.parametrize("loop_nest", [0, 1, 2])
def test_frames_equal_libzstd(zs, oracle, loop_nest):
    """both forms of the parser: the micro-step loop the kernel runs (0) and the plain loop nest (1)"""
    bad = []
    for i, p in enumerate(ZC.corpus(oracle, 2024, 160)):
        if _my_frame(zs, p, loop_nest) != ZC.ref_frame(p):
            bad.append((i, len(p)))
    assert not bad, bad


@pytest.mark.parametrize("g", [1, 2, 3])
def test_group_parser_frames_equal_libzstd(zs, oracle, g):
    """zs_opt_grp.h: g lanes per frame on consecutive positions, recorded tree walks committed after an in-order validation
    (the host build runs every segment of a trip for lane 0..g-1 in turn); g = 1 is the same code with the leader alone"""
    bad = []
    for i, p in enumerate(ZC.corpus(oracle, 777 + g, 120, max_len=16384)):
        if _my_frame_grp(zs, p, g) != ZC.ref_frame(p):
            bad.append((i, len(p)))
    rng = np.random.default_rng(g)
    for n_samp in (3, 11, 25, 31):  # the packs of a Close(): n_samp deltas of one group
        p = ZC.delta_pack(oracle, rng, n_samp, 60000, 1e-3)[:16384]
        if _my_frame_grp(zs, p, g) != ZC.ref_frame(p):
            bad.append(("pack", n_samp, len(p)))
    # anomalies on purpose: runs (equal hash buckets in a row, matches running to the end of the block), long repeats
    # (immediate encoding, skipped areas), a pattern with a period below minMatch
    for p in (b"A" * 9000, b"AB" * 4000, b"ABC" * 3000, (b"0,1234.C" * 40 + b"!!!!!!!!!!!!" ) * 30, bytes(rng.integers(65, 69, 6000, dtype=np.uint8)) * 2,
              b"".join(bytes([65 + (i * 7) % 23]) * (1 + i % 9) for i in range(3000))):
        p = p[:16384]
        if _my_frame_grp(zs, p, g) != ZC.ref_frame(p):
            bad.append(("special", len(p)))
    assert not bad, bad


@pytest.mark.parametrize("g", [2, 3])
def test_group_parser_takes_inputs_up_to_one_block(zs, oracle, g):
    """beyond the btultra2 class (> 16 KiB: btultra, one pass, deeper trees, tree slots and indices beyond 16 bits) the walks are
    recorded with two words per store (zs::grpWide); full packs of a collection (100 deltas) are of that size"""
    rng = np.random.default_rng(60 + g)
    inputs = [p for p in ZC.corpus(oracle, 4000 + g, 96) if len(p) > 16384]
    inputs += [ZC.delta_pack(oracle, rng, n, 60000, 1e-3) for n in (40, 100)] + [b"A" * 70000, b"ABC" * 30000, b"x" * 131072]
    assert len(inputs) > 20
    bad = [(i, len(p)) for i, p in enumerate(inputs) if _my_frame_grp(zs, p, g) != ZC.ref_frame(p)]
    assert not bad, bad


def test_parser_sequences_equal_generate_sequences(zs, oracle):
    rng = np.random.default_rng(7)
    for n_samp, seg in ((1, 60000), (20, 60000), (100, 60000), (100, 30000), (7, 5000)):
        p = ZC.delta_pack(oracle, rng, n_samp, seg, 1e-3)
        assert _my_sequences(zs, p) == ZC.ref_sequences(p), (n_samp, seg, len(p))


def test_level17_parameters_equal_getcparams():
    from agc_amd import capi
    L = capi.load()
    out = (C.c_uint32 * 7)()
    # (0 means "unknown" to the public ZSTD_getCParams; the one-shot path of an empty input is covered by the frame test)
    sizes = list(range(1, 3000)) + list(range(3000, 140000, 97)) + [16383, 16384, 16385, 131071, 131072, 131073, 262144, 262145, 1 << 20]
    for n in sizes:
        assert L.agc_hip_zstd17_cparams(n, out) == 0
        assert list(out) == ZC.ref_cparams(n), n
    assert L.agc_hip_zstd17_max_input() == 131072


def test_level_parameters_equal_getcparams():
    """the rows of levels 13 (tuple-packed references) and 19 (repetitive references) as well: segment.h:172-255"""
    from agc_amd import capi
    L = capi.load()
    out = (C.c_uint32 * 7)()
    sizes = list(range(1, 600)) + list(range(600, 140000, 389)) + [16383, 16384, 16385, 131071, 131072, 131073, 262144, 262145, 1 << 20, 1 << 24]
    for level in (13, 17, 19):
        for n in sizes:
            assert L.agc_hip_zstd_cparams(level, n, out) == 0
            assert list(out) == ZC.ref_cparams(n, level), (level, n)
    assert L.agc_hip_zstd_cparams(12, 1000, out) != 0


def _my_frame_level(H, data, level, mode):
    n = len(data)
    cp = np.array(ZC.ref_cparams(n, level), np.uint32)
    out = np.zeros(n + 64, np.uint8)
    if mode >= 10:
        k = H.zs_host_compress_grp(bytes(data), n, cp.ctypes.data, out.ctypes.data, mode - 10)
    else:
        k = H.zs_host_compress2(bytes(data), n, cp.ctypes.data, out.ctypes.data, mode)
    return out[:k].tobytes()


def reference_like_inputs(rng):
    """what CSegment::add_to_archive_tuples / store_in_archive(ref) compress: ACGT segments packed 4 symbols per byte (level 13),
    segments with N (3 per byte), and repetitive segments as raw symbols (level 19)"""
    from agc_amd import synth
    out = []
    for n_sym in (9000, 40000, 61000, 65000, 66000, 120000, 400000):
        c = np.asarray(synth.random_seq(rng, n_sym), dtype=np.uint8)
        c4 = c[:len(c) // 4 * 4].reshape(-1, 4)
        out.append((13, (c4[:, 0] * 64 + c4[:, 1] * 16 + c4[:, 2] * 4 + c4[:, 3]).astype(np.uint8).tobytes()))
        c[rng.integers(0, n_sym, 5)] = 4
        c3 = c[:len(c) // 3 * 3].reshape(-1, 3)
        out.append((13, (c3[:, 0] * 36 + c3[:, 1] * 6 + c3[:, 2]).astype(np.uint8).tobytes()))
    for n_sym in (5000, 16384, 16385, 60000, 131072):
        unit = np.asarray(synth.random_seq(rng, 41), dtype=np.uint8)
        s = np.tile(unit, n_sym // 41 + 1)[:n_sym].copy()
        idx = rng.integers(0, n_sym, n_sym // 300)
        s[idx] = rng.integers(0, 4, len(idx))
        out.append((19, s.tobytes()))
    return [(lv, d) for lv, d in out if len(d) <= 131072]


def test_frames_of_levels_13_and_19_equal_libzstd(zs):
    """the same headers at the other two levels of the archive: level 13 is btultra (minMatch 3) up to 16 KiB and btopt (minMatch 4,
    bit-weight prices) above, level 19 btultra2 (two passes) at every size -- loop nest, micro-step loop with the device's table
    split, and the lane-group parser where it is eligible"""
    rng = np.random.default_rng(1319)
    bad = []
    for level, data in reference_like_inputs(rng):
        want = ZC.ref_frame(data, level)
        for mode in (0, 2, 13):
            if _my_frame_level(zs, data, level, mode) != want:
                bad.append((level, len(data), mode))
    assert not bad, bad


def test_code_functions_equal_the_format_tables(zs):
    """LL_bits / ML_bits / LL_Code / ML_Code of zstd_internal.h (RFC 8878 3.1.1.3.2.1.1) against the arithmetic the kernels use"""
    LL_bits = [0] * 16 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    ML_bits = [0] * 32 + [1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    LL_Code = list(range(16)) + [16, 16, 17, 17, 18, 18, 19, 19] + [20] * 4 + [21] * 4 + [22] * 8 + [23] * 8 + [24] * 16
    ML_Code = list(range(32)) + [32, 32, 33, 33, 34, 34, 35, 35] + [36] * 4 + [37] * 4 + [38] * 8 + [39] * 8 + [40] * 16 + [41] * 16 + [42] * 32
    a, b, c, d = (C.c_uint32 * 36)(), (C.c_uint32 * 53)(), (C.c_uint32 * 64)(), (C.c_uint32 * 128)()
    zs.zs_host_code_tables(a, b, c, d)
    assert list(a) == LL_bits and list(b) == ML_bits and list(c) == LL_Code and list(d) == ML_Code
