"""ctypes binding of the read side (libagc_read.so, include/agc_read.h).

`CAGCFile` mirrors the class the reference exports to Python (src/py_agc_api/py_agc_api.cpp:28-84: Open, Close,
NSample, GetReferenceSample, NCtg, ListSample, ListCtg, GetCtgLen, GetCtgSeq with and without a sample name), so
src/py_agc_api/py_agc_test.py reads the same against this module (`StringVector` is a plain list here).
Host code only -- no torch, no HIP."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libagc_read.so")
SYMBOLS = ["agc_open", "agc_close", "agc_get_ctg_len", "agc_get_ctg_seq", "agc_n_sample", "agc_n_ctg", "agc_reference_sample",
           "agc_list_sample", "agc_list_ctg", "agc_list_destroy", "agc_string_destroy", "agc_get_sample_fasta", "agc_get_params"]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m agc_amd.build`")
    L = C.CDLL(LIB_PATH)
    vp, cp, ci = C.c_void_p, C.c_char_p, C.c_int
    L.agc_open.restype = vp
    L.agc_open.argtypes = [cp, ci]
    L.agc_close.argtypes = [vp]
    L.agc_get_ctg_len.argtypes = [vp, cp, cp]
    L.agc_get_ctg_seq.argtypes = [vp, cp, cp, ci, ci, cp]
    L.agc_n_sample.argtypes = [vp]
    L.agc_n_ctg.argtypes = [vp, cp]
    L.agc_reference_sample.restype = vp
    L.agc_reference_sample.argtypes = [vp]
    L.agc_list_sample.restype = C.POINTER(vp)
    L.agc_list_sample.argtypes = [vp, C.POINTER(ci)]
    L.agc_list_ctg.restype = C.POINTER(vp)
    L.agc_list_ctg.argtypes = [vp, cp, C.POINTER(ci)]
    L.agc_list_destroy.argtypes = [C.POINTER(vp)]
    L.agc_string_destroy.argtypes = [vp]
    L.agc_get_sample_fasta.restype = vp
    L.agc_get_sample_fasta.argtypes = [vp, cp, ci, C.POINTER(C.c_longlong)]
    L.agc_get_params.argtypes = [vp] + [C.POINTER(C.c_uint)] * 4
    _lib = L
    return L


class StringVector(list):
    """stand-in for py_agc_api.StringVector (py_agc_api.cpp:18-26)"""


class CAGCFile:
    def __init__(self):
        self.L = load()
        self.h = None

    def Open(self, file_name, prefetching=True):
        if self.h:
            return False
        self.h = self.L.agc_open(os.fsencode(file_name), 1 if prefetching else 0)
        return bool(self.h)

    def Close(self):
        if not self.h:
            return False
        r = self.L.agc_close(self.h)
        self.h = None
        return r == 0

    def __del__(self):
        if getattr(self, "h", None):
            self.Close()

    def NSample(self):
        return self.L.agc_n_sample(self.h) if self.h else -1

    def NCtg(self, sample):
        return self.L.agc_n_ctg(self.h, sample.encode()) if self.h else -1

    def GetReferenceSample(self):
        if not self.h:
            return ""
        p = self.L.agc_reference_sample(self.h)
        if not p:
            return ""
        s = C.string_at(p).decode()
        self.L.agc_string_destroy(p)
        return s

    def _list(self, arr, n, out):
        names = [C.string_at(arr[i]).decode() for i in range(n)] if arr else []
        if arr:
            self.L.agc_list_destroy(arr)
        if out is not None:
            del out[:]
            out.extend(names)
            return 0
        return names

    def ListSample(self, out=None):
        if not self.h:
            return -1
        n = C.c_int(0)
        return self._list(self.L.agc_list_sample(self.h, C.byref(n)), n.value, out)

    def ListCtg(self, sample, out=None):
        if not self.h:
            return -1
        n = C.c_int(0)
        return self._list(self.L.agc_list_ctg(self.h, sample.encode(), C.byref(n)), n.value, out)

    def GetCtgLen(self, sample, name):
        if not self.h:
            return -1
        return self.L.agc_get_ctg_len(self.h, sample.encode() if sample else None, name.encode())

    def GetCtgSeq(self, *args):
        """GetCtgSeq(sample, name, start, end) or GetCtgSeq(name, start, end); [start, end] inclusive, -1/-1 = all"""
        if len(args) == 3:
            sample, (name, start, end) = "", args
        else:
            sample, name, start, end = args
        if not self.h:
            return ""
        n = self.GetCtgLen(sample, name)
        if n < 0:
            return ""
        buf = C.create_string_buffer(n + 1)
        r = self.L.agc_get_ctg_seq(self.h, sample.encode() if sample else None, name.encode(), start, end, buf)
        return buf.raw[:r].decode() if r >= 0 else ""

    # extensions
    def GetSampleFasta(self, sample, line_length=80):
        n = C.c_longlong(0)
        p = self.L.agc_get_sample_fasta(self.h, sample.encode(), line_length, C.byref(n))
        if not p:
            return None
        s = C.string_at(p, n.value)
        self.L.agc_string_destroy(p)
        return s

    def GetParams(self):
        v = [C.c_uint(0) for _ in range(4)]
        if self.L.agc_get_params(self.h, *[C.byref(x) for x in v]) != 0:
            return None
        return dict(zip(("k", "min_match_len", "pack_cardinality", "segment_size"), (x.value for x in v)))
