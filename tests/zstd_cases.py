"""Inputs for the zstd level-17 parity tests and the bindings of the libzstd they are compared with.
The delta packs are what CSegment::store_in_archive(pack) hands to ZSTD_compressCCtx (src/common/segment.h:258-280):
LZ-diff deltas of one group's segments, each followed by 0xFF."""
import ctypes as C
import os

import numpy as np

ZSTD_PATHS = ["/opt/conda/lib/libzstd.so.1", "libzstd.so.1"]


class CParams(C.Structure):
    _fields_ = [("windowLog", C.c_uint), ("chainLog", C.c_uint), ("hashLog", C.c_uint), ("searchLog", C.c_uint), ("minMatch", C.c_uint),
                ("targetLength", C.c_uint), ("strategy", C.c_int)]


class Seq(C.Structure):
    _fields_ = [("offset", C.c_uint), ("litLength", C.c_uint), ("matchLength", C.c_uint), ("rep", C.c_uint)]


_z = None


def libzstd():
    """the image's libzstd (1.4.9: the version the archives are pinned against)"""
    global _z
    if _z is None:
        for p in ZSTD_PATHS:
            try:
                _z = C.CDLL(p, mode=os.RTLD_LOCAL | os.RTLD_DEEPBIND)
                break
            except OSError:
                continue
        Z = _z
        Z.ZSTD_versionNumber.restype = C.c_uint
        Z.ZSTD_getCParams.restype = CParams
        Z.ZSTD_getCParams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t]
        Z.ZSTD_createCCtx.restype = C.c_void_p
        Z.ZSTD_compressBound.restype = C.c_size_t
        Z.ZSTD_compressBound.argtypes = [C.c_size_t]
        Z.ZSTD_compressCCtx.restype = C.c_size_t
        Z.ZSTD_compressCCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        Z.ZSTD_generateSequences.restype = C.c_size_t
        Z.ZSTD_generateSequences.argtypes = [C.c_void_p, C.POINTER(Seq), C.c_size_t, C.c_void_p, C.c_size_t]
        Z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        Z._cctx = Z.ZSTD_createCCtx()
        Z._cctx_seq = Z.ZSTD_createCCtx()
        Z.ZSTD_CCtx_setParameter(Z._cctx_seq, 100, 17)  # ZSTD_c_compressionLevel
    return _z


def ref_frame(data, level=17):
    Z = libzstd()
    n = len(data)
    cap = Z.ZSTD_compressBound(n)
    out = C.create_string_buffer(cap)
    k = Z.ZSTD_compressCCtx(Z._cctx, out, cap, bytes(data), n, level)
    return out.raw[:k]


def ref_sequences(data):
    """ZSTD_generateSequences at level 17: (offset, litLength, matchLength, rep) incl. the block delimiter"""
    Z = libzstd()
    n = len(data)
    buf = (Seq * (n // 3 + 16))()
    k = Z.ZSTD_generateSequences(Z._cctx_seq, buf, len(buf), bytes(data), n)
    return [(s.offset, s.litLength, s.matchLength, s.rep) for s in buf[:k]]


def ref_cparams(n, level=17):
    p = libzstd().ZSTD_getCParams(level, n, 0)
    return [p.windowLog, p.chainLog, p.hashLog, p.searchLog, p.minMatch, p.targetLength, p.strategy]


def delta_pack(oracle, rng, n_samples, seg_len, d, mml=15):
    """one group's delta pack: n_samples mutated copies of a random reference segment, LZ-diff encoded, 0xFF-separated"""
    from agc_amd import synth
    ref = synth.random_seq(rng, seg_len)
    z = oracle.LZ(ref, mml)
    return b"".join(z.encode(synth.mutate(rng, ref, d)).tobytes() + b"\xff" for _ in range(n_samples))


def corpus(oracle, seed, count, max_len=131072):
    """a mixed corpus: delta packs of many shapes, raw packs (symbol codes), text-like, skewed, mutated repeats, edge sizes"""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(count):
        kind = it % 8
        if kind < 4:
            n_samp = int(rng.integers(1, 101))
            seg = int(rng.choice([5000, 20000, 60000, 120000]))
            d = float(rng.choice([1e-4, 1e-3, 3e-3, 1e-2]))
            if n_samp * seg * d * 14 > 120000:
                n_samp = max(1, int(120000 / (seg * d * 14)))
            p = delta_pack(oracle, rng, n_samp, seg, d)
        elif kind == 4:
            p = b"".join(bytes(rng.integers(0, 4, int(rng.integers(10, 20000)), dtype=np.uint8)) + b"\xff" for _ in range(int(rng.integers(1, 6))))
        elif kind == 5:
            words = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(int(rng.integers(5, 200)))]
            p = b" ".join(words[int(rng.integers(len(words)))] for _ in range(int(rng.integers(10, 15000))))
        elif kind == 6:
            k = int(rng.integers(2, 40))
            probs = rng.dirichlet(np.ones(k) * 0.3)
            p = bytes(rng.choice(k, size=int(rng.integers(7, 60000)), p=probs).astype(np.uint8) + 32)
        else:
            blk = bytes(rng.integers(0, 256, int(rng.integers(20, 3000)), dtype=np.uint8))
            arr = np.frombuffer(blk * int(rng.integers(2, 40)), np.uint8).copy()
            m = rng.random(arr.size) < 0.01
            arr[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
            p = arr.tobytes()
        out.append(p[:max_len])
    base = out[0] if out else b"x" * 500
    for n in (0, 1, 2, 6, 7, 8, 9, 16, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025):
        out.append((base * (n // max(len(base), 1) + 1))[:n])
    out.append(bytes(rng.integers(0, 256, 5000, dtype=np.uint8)))   # incompressible -> raw block
    out.append(b"A" * 10000)                                          # rle literals
    out.append(b"ACGT" * 3000)
    return out
