"""ctypes front-end of oracle/agc_oracle.c (CPU restatement) and, when it was
prebuilt in the authoring container, of oracle/_ref/libagcref.so (the
reference's own LZ-diff / k-mer code compiled in place).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from agc_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libagc_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libagcref.so")
REF_AGC = os.path.join(_HERE, "_ref", "agc")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile the C restatement (and the reference, when its tree is here)."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "agc_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, os.path.join(_HERE, "libagc_oracle.so")])
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(_REF) or not os.path.exists(REF_AGC)):
        subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref"])


def _p(a, t=u8p):
    return a.ctypes.data_as(t)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.agco_preprocess.restype = C.c_size_t
        L.agco_preprocess.argtypes = [u8p, C.c_size_t, u8p]
        L.agco_rev_comp.argtypes = [u8p, C.c_size_t, u8p]
        L.agco_murmur64.restype = C.c_uint64
        L.agco_murmur64.argtypes = [C.c_uint64]
        L.agco_scan_contig.restype = C.c_size_t
        L.agco_scan_contig.argtypes = [u8p, C.c_size_t, C.c_uint32, u64p, C.c_size_t, C.c_size_t,
                                       u64p, u64p, u64p, u64p, u8p, u64p, u64p, u8p]
        L.agco_enumerate_kmers.restype = C.c_size_t
        L.agco_enumerate_kmers.argtypes = [u8p, C.c_size_t, C.c_uint32, u64p]
        L.agco_sort_keep_singletons.restype = C.c_size_t
        L.agco_sort_keep_singletons.argtypes = [u64p, C.c_size_t]
        L.agco_find_splitters_in_contig.restype = C.c_size_t
        L.agco_find_splitters_in_contig.argtypes = [u8p, C.c_size_t, C.c_uint32, C.c_uint64, u64p, C.c_size_t, u64p]
        L.agco_lz_create.restype = C.c_void_p
        L.agco_lz_create.argtypes = [u8p, C.c_uint32, C.c_uint32]
        L.agco_lz_free.argtypes = [C.c_void_p]
        L.agco_lz_index.restype = C.c_uint64
        L.agco_lz_index.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.agco_lz_encode.restype = C.c_size_t
        L.agco_lz_encode.argtypes = [C.c_void_p, u8p, C.c_uint32, u8p]
        L.agco_lz_estimate.restype = C.c_uint32
        L.agco_lz_estimate.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_uint32, u32p]
        L.agco_lz_cost_vector.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_int, u32p]
        L.agco_lz_decode.restype = C.c_size_t
        L.agco_lz_decode.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p, C.c_size_t, u8p, C.c_size_t]
        L.agco_ref_is_repetitive.restype = C.c_int
        L.agco_ref_is_repetitive.argtypes = [u8p, C.c_size_t]
        L.agco_ref_lag_counts.argtypes = [u8p, C.c_size_t, u32p, u32p]
        L.agco_bytes2tuples.restype = C.c_size_t
        L.agco_bytes2tuples.argtypes = [u8p, C.c_size_t, u8p]
        _lib = L
    return _lib


# --------------------------------------------------------------------------
def preprocess(raw):
    raw = _u8(raw)
    out = np.empty(raw.size, np.uint8)
    n = lib().agco_preprocess(_p(raw), raw.size, _p(out))
    return out[:n].copy()


def rev_comp(seq):
    seq = _u8(seq)
    out = np.empty_like(seq)
    lib().agco_rev_comp(_p(seq), seq.size, _p(out))
    return out


def scan_contig(ctg, k, splitters_sorted):
    """-> dict of arrays: start, len, front_dir, front_rc, front_full, back_*"""
    ctg = _u8(ctg)
    spl = np.ascontiguousarray(splitters_sorted, dtype=np.uint64)
    cap = 64
    while True:
        a = {n: np.zeros(cap, np.uint64) for n in ("start", "len", "front_dir", "front_rc", "back_dir", "back_rc")}
        ff = np.zeros(cap, np.uint8)
        bf = np.zeros(cap, np.uint8)
        n = lib().agco_scan_contig(_p(ctg), ctg.size, k, _p(spl, u64p), spl.size, cap,
                                   _p(a["start"], u64p), _p(a["len"], u64p),
                                   _p(a["front_dir"], u64p), _p(a["front_rc"], u64p), _p(ff),
                                   _p(a["back_dir"], u64p), _p(a["back_rc"], u64p), _p(bf))
        if n <= cap:
            out = {k_: v[:n].copy() for k_, v in a.items()}
            out["front_full"] = ff[:n].copy()
            out["back_full"] = bf[:n].copy()
            return out
        cap = n


def determine_splitters(ref_contigs, k, segment_size):
    """Reference preprocessing (agc_compressor.cpp:428-563): sorted unique splitter k-mers."""
    L = lib()
    parts = []
    for ctg in ref_contigs:
        ctg = _u8(ctg)
        buf = np.empty(max(ctg.size, 1), np.uint64)
        n = L.agco_enumerate_kmers(_p(ctg), ctg.size, k, _p(buf, u64p))
        parts.append(buf[:n])
    allk = np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)
    n = L.agco_sort_keep_singletons(_p(allk, u64p), allk.size)
    sing = np.ascontiguousarray(allk[:n])
    spl = []
    for ctg in ref_contigs:
        ctg = _u8(ctg)
        out = np.empty(ctg.size // max(1, segment_size) + 4, np.uint64)
        m = L.agco_find_splitters_in_contig(_p(ctg), ctg.size, k, segment_size, _p(sing, u64p), sing.size, _p(out, u64p))
        spl.append(out[:m])
    return np.unique(np.concatenate(spl)) if spl else np.zeros(0, np.uint64)


class LZ:
    """CLZDiff_V2 restatement bound to one reference sequence."""

    def __init__(self, ref, min_match_len):
        self.ref = _u8(ref)
        self.mml = int(min_match_len)
        self.h = lib().agco_lz_create(_p(self.ref), self.ref.size, self.mml)

    def __del__(self):
        if getattr(self, "h", None):
            lib().agco_lz_free(self.h)
            self.h = None

    def index(self):
        is16 = C.c_int()
        tab = C.c_void_p()
        n = lib().agco_lz_index(self.h, C.byref(is16), C.byref(tab))
        dt = np.uint16 if is16.value else np.uint32
        arr = np.ctypeslib.as_array(C.cast(tab, C.POINTER(C.c_uint16 if is16.value else C.c_uint32)), shape=(n,))
        return arr.astype(dt).copy()

    def encode(self, text):
        text = _u8(text)
        out = np.empty(text.size + 5 * text.size // 16 + 64, np.uint8)
        n = lib().agco_lz_encode(self.h, _p(text), text.size, _p(out))
        return out[:n].copy()

    def estimate(self, text, bound=0xFFFFFFFF, want_peak=False):
        text = _u8(text)
        peak = C.c_uint32()
        r = lib().agco_lz_estimate(self.h, _p(text), text.size, bound, C.byref(peak))
        return (r, peak.value) if want_peak else r

    def cost_vector(self, text, prefix_costs):
        text = _u8(text)
        out = np.zeros(text.size, np.uint32)
        lib().agco_lz_cost_vector(self.h, _p(text), text.size, int(bool(prefix_costs)), _p(out, u32p))
        return out

    def decode(self, enc, cap):
        enc = _u8(enc)
        out = np.empty(cap, np.uint8)
        n = lib().agco_lz_decode(_p(self.ref), self.ref.size, self.mml, _p(enc), enc.size, _p(out), cap)
        return out[:min(n, cap)].copy(), n


def ref_is_repetitive(data):
    data = _u8(data)
    return bool(lib().agco_ref_is_repetitive(_p(data), data.size))


def ref_lag_counts(data):
    data = _u8(data)
    cnt = np.zeros(28, np.uint32)
    cur = np.zeros(28, np.uint32)
    lib().agco_ref_lag_counts(_p(data), data.size, _p(cnt, u32p), _p(cur, u32p))
    return cnt, cur


def bytes2tuples(data):
    data = _u8(data)
    out = np.empty(data.size + 2, np.uint8)
    n = lib().agco_bytes2tuples(_p(data), data.size, _p(out))
    return out[:n].copy()


# --------------------------------------------------------------------------
# The reference itself (only where oracle/_ref was prebuilt).
_ref = None


def have_ref():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(_REF)
        R.ref_lz_create.restype = C.c_void_p
        R.ref_lz_create.argtypes = [u8p, C.c_uint32, C.c_uint32]
        R.ref_lz_free.argtypes = [C.c_void_p]
        R.ref_lz_encode.restype = C.c_size_t
        R.ref_lz_encode.argtypes = [C.c_void_p, u8p, C.c_uint32, u8p, C.c_size_t]
        R.ref_lz_estimate.restype = C.c_uint64
        R.ref_lz_estimate.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_uint32]
        R.ref_lz_cost_vector.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_int, u32p]
        R.ref_lz_decode.restype = C.c_size_t
        R.ref_lz_decode.argtypes = [C.c_uint32, u8p, C.c_uint32, u8p, C.c_size_t, u8p, C.c_size_t]
        R.ref_scan_hits.restype = C.c_size_t
        R.ref_scan_hits.argtypes = [u8p, C.c_size_t, C.c_uint32, u64p, C.c_size_t, C.c_size_t, u64p, u64p, u64p]
        _ref = R
    return _ref


class RefLZ:
    """The reference's CLZDiff_V2 (src/common/lz_diff.h:375-436) via ref_harness.cpp."""

    def __init__(self, refseq, min_match_len):
        self.ref = _u8(refseq)
        self.mml = int(min_match_len)
        self.h = ref().ref_lz_create(_p(self.ref), self.ref.size, self.mml)

    def __del__(self):
        if getattr(self, "h", None):
            ref().ref_lz_free(self.h)
            self.h = None

    def encode(self, text):
        text = _u8(text)
        cap = 2 * text.size + 64
        out = np.empty(cap, np.uint8)
        n = ref().ref_lz_encode(self.h, _p(text), text.size, _p(out), cap)
        return out[:n].copy()

    def estimate(self, text, bound=0xFFFFFFFF):
        text = _u8(text)
        return int(ref().ref_lz_estimate(self.h, _p(text), text.size, bound))

    def cost_vector(self, text, prefix_costs):
        text = _u8(text)
        out = np.zeros(text.size, np.uint32)
        ref().ref_lz_cost_vector(self.h, _p(text), text.size, int(bool(prefix_costs)), _p(out, u32p))
        return out


def ref_scan_hits(ctg, k, splitters):
    ctg = _u8(ctg)
    spl = np.ascontiguousarray(splitters, dtype=np.uint64)
    cap = max(64, ctg.size // 16)
    pos = np.zeros(cap, np.uint64)
    d = np.zeros(cap, np.uint64)
    r = np.zeros(cap, np.uint64)
    n = ref().ref_scan_hits(_p(ctg), ctg.size, k, _p(spl, u64p), spl.size, cap, _p(pos, u64p), _p(d, u64p), _p(r, u64p))
    assert n <= cap
    return pos[:n].copy(), d[:n].copy(), r[:n].copy()
