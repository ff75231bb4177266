"""ctypes binding of include/agc_hip.h (libagc_hip.so).

This is plumbing for tests and bench.py: numpy arrays for host buffers, raw
device pointers (e.g. torch tensors' data_ptr()) for HBM buffers.  There is no
fallback: if the library is missing or no HIP device is present, calls raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libagc_hip.so")

OK, ENODEV, EINVAL, ENOMEM, ECAP, ENOREF = 0, -1, -2, -3, -4, -5
K_SCAN, K_INDEX, K_ENCODE, K_ESTIMATE, K_COSTVEC, K_REVCOMP, K_PREPROCESS, K_REFSTORE = range(8)
K_NAMES = ["scan", "index", "encode", "estimate", "costvec", "revcomp", "preprocess", "refstore", "zstd", "filter", "segments", "pack"]

# every symbol include/agc_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "agc_hip_create", "agc_hip_destroy", "agc_hip_last_error", "agc_hip_abi_version", "agc_hip_sync",
    "agc_hip_timing_enable", "agc_hip_timing_reset", "agc_hip_timing_get",
    "agc_hip_sample_buffer", "agc_hip_copy_to_device",
    "agc_hip_preprocess_dev", "agc_hip_preprocess",
    "agc_hip_splitters_set", "agc_hip_splitters_insert", "agc_hip_splitters_count",
    "agc_hip_determine_splitters_dev",
    "agc_hip_scan_contigs_dev", "agc_hip_scan_contigs",
    "agc_hip_ref_register", "agc_hip_ref_register_batch_dev", "agc_hip_ref_get", "agc_hip_ref_index_get",
    "agc_hip_lz_encode_batch_dev", "agc_hip_lz_encode_batch", "agc_hip_lz_encode_begin_dev", "agc_hip_lz_encode_end", "agc_hip_lz_encode_pending",
    "agc_hip_lz_encode_begin_packed_on", "agc_hip_lz_encode_end_on", "agc_hip_lz_encode_pending_on", "agc_hip_lz_encode_drop_on",
    "agc_hip_host_alloc", "agc_hip_host_free",
    "agc_hip_lz_estimate_batch_dev", "agc_hip_lz_estimate_batch",
    "agc_hip_lz_cost_vector_batch_dev", "agc_hip_lz_cost_vector_batch",
    "agc_hip_lz_split_point_batch_dev", "agc_hip_fetch_slices_dev",
    "agc_hip_ref_lag_counts_dev",
    "agc_hip_zstd17_max_input", "agc_hip_zstd17_resident_frames", "agc_hip_zstd17_batch", "agc_hip_zstd17_batch_dev", "agc_hip_zstd17_background", "agc_hip_zstd17_cparams", "agc_hip_zstd_batch", "agc_hip_zstd_cparams",
    "agc_hip_packed_words_bytes", "agc_hip_packed_index_bytes", "agc_hip_pack_dev", "agc_hip_expand_dev", "agc_hip_scan_packed_dev",
    "agc_hip_prefetch_packed_dev", "agc_hip_scan_prefetched", "agc_hip_sample_pack", "agc_hip_sample_pack_fasta", "agc_hip_ref_store_begin_packed", "agc_hip_ref_store_end",
    "agc_hip_ref_register_batch_packed", "agc_hip_lz_encode_batch_packed", "agc_hip_lz_encode_begin_packed", "agc_hip_lz_estimate_batch_packed",
    "agc_hip_lz_cost_vector_batch_packed", "agc_hip_lz_split_point_batch_packed", "agc_hip_fetch_slices_packed", "agc_hip_ref_lag_counts_packed",
    "agc_hip_pack_fasta_begin", "agc_hip_pack_fasta_end", "agc_hip_pack_fasta_dev",
    "agc_hip_group_hash", "agc_hip_group_map_set", "agc_hip_group_map_update", "agc_hip_segments_packed", "agc_hip_segments_encode_known",
]

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p


class Packed(C.Structure):
    """agc_hip_packed (include/agc_hip.h): a sample in the 2-bit HBM layout"""
    _fields_ = [("d_words", C.c_void_p), ("d_esc_index", C.c_void_p), ("d_esc_bytes", C.c_void_p), ("n_symbols", C.c_uint64)]


class GroupSlot(C.Structure):
    """agc_hip_group_slot: one slot of the (k-mer 1, k-mer 2) -> group table"""
    _fields_ = [("k1", C.c_uint64), ("k2", C.c_uint64), ("gid", C.c_int32), ("used", C.c_uint32)]


class Segment(C.Structure):
    """agc_hip_segment"""
    _fields_ = [("start", C.c_uint64), ("front_dir", C.c_uint64), ("front_rc", C.c_uint64), ("back_dir", C.c_uint64), ("back_rc", C.c_uint64),
                ("ctg", C.c_uint32), ("len", C.c_uint32), ("map_gid", C.c_int32), ("front_full", C.c_uint8), ("back_full", C.c_uint8),
                ("store_rc", C.c_uint8), ("encoded", C.c_uint8)]


SEGMENT_DTYPE = np.dtype([("start", "<u8"), ("front_dir", "<u8"), ("front_rc", "<u8"), ("back_dir", "<u8"), ("back_rc", "<u8"), ("ctg", "<u4"), ("len", "<u4"),
                          ("map_gid", "<i4"), ("front_full", "u1"), ("back_full", "u1"), ("store_rc", "u1"), ("encoded", "u1")])
GROUP_SLOT_DTYPE = np.dtype([("k1", "<u8"), ("k2", "<u8"), ("gid", "<i4"), ("used", "<u4")])


class AgcHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"agc_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """dlopen libagc_hip.so; raises if it was not built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; importing torch
    # first makes libagc_hip.so bind to that already-loaded runtime (same SONAME) instead of
    # pulling in a second copy from /opt/rocm, after which torch would see no GPUs.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m agc_amd.build` (hipcc, gfx950) first")
    L = C.CDLL(os.environ.get("AGC_HIP_LIB", LIB_PATH))  # (AGC_HIP_LIB: a kernel-variant build for scripts/ probes)
    L.agc_hip_create.argtypes = [C.POINTER(vp), C.c_int]
    L.agc_hip_destroy.argtypes = [vp]
    L.agc_hip_destroy.restype = None
    L.agc_hip_last_error.argtypes = [vp]
    L.agc_hip_last_error.restype = C.c_char_p
    L.agc_hip_abi_version.restype = C.c_uint32
    L.agc_hip_sync.argtypes = [vp]
    L.agc_hip_timing_enable.argtypes = [vp, C.c_int]
    L.agc_hip_timing_reset.argtypes = [vp]
    L.agc_hip_timing_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), u64p]
    L.agc_hip_sample_buffer.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.agc_hip_copy_to_device.argtypes = [vp, vp, u8p, C.c_uint64]
    L.agc_hip_preprocess_dev.argtypes = [vp, vp, C.c_uint64, vp, u64p]
    L.agc_hip_splitters_set.argtypes = [vp, u64p, C.c_uint64]
    L.agc_hip_splitters_insert.argtypes = [vp, u64p, C.c_uint64]
    L.agc_hip_splitters_count.argtypes = [vp]
    L.agc_hip_splitters_count.restype = C.c_uint64
    L.agc_hip_determine_splitters_dev.argtypes = [vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, u64p, u64p,
                                                  C.c_uint64, u64p, u64p]
    scan_tail = [u64p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, u32p, u64p, u64p, u64p]
    L.agc_hip_scan_contigs_dev.argtypes = [vp, vp] + scan_tail
    L.agc_hip_scan_contigs.argtypes = [vp, u8p] + scan_tail
    L.agc_hip_ref_register.argtypes = [vp, C.c_uint32, u8p, C.c_uint32, C.c_uint32]
    L.agc_hip_ref_register_batch_dev.argtypes = [vp, C.c_uint32, u32p, vp, u64p, u32p, u8p, C.c_uint32]
    L.agc_hip_ref_get.argtypes = [vp, C.c_uint32, u8p, C.c_uint32, u32p]
    L.agc_hip_ref_index_get.argtypes = [vp, C.c_uint32, u32p, C.c_uint64, u64p, C.POINTER(C.c_int)]
    L.agc_hip_lz_encode_batch_dev.argtypes = [vp, C.c_uint32, u32p, vp, u64p, u32p, u8p, u8p, C.c_uint64, u64p]
    L.agc_hip_lz_encode_batch.argtypes = [vp, C.c_uint32, u32p, u8p, u64p, u32p, u8p, u8p, C.c_uint64, u64p]
    L.agc_hip_lz_encode_begin_dev.argtypes = [vp, C.c_uint32, u32p, vp, u64p, u32p, u8p]
    L.agc_hip_lz_encode_end.argtypes = [vp, u8p, C.c_uint64, u64p]
    L.agc_hip_lz_encode_pending.argtypes = [vp, u32p]
    L.agc_hip_host_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.agc_hip_host_free.argtypes = [vp, vp]
    L.agc_hip_lz_estimate_batch_dev.argtypes = [vp, C.c_uint32, u32p, vp, u64p, u32p, u8p, u32p, u32p]
    L.agc_hip_lz_estimate_batch.argtypes = [vp, C.c_uint32, u32p, u8p, u64p, u32p, u8p, u32p, u32p]
    L.agc_hip_lz_cost_vector_batch_dev.argtypes = [vp, C.c_uint32, u32p, vp, u64p, u32p, u8p, u8p, u32p]
    L.agc_hip_lz_cost_vector_batch.argtypes = [vp, C.c_uint32, u32p, u8p, u64p, u32p, u8p, u8p, u32p]
    L.agc_hip_lz_split_point_batch_dev.argtypes = [vp, C.c_uint32, u32p, u32p, vp, u64p, u32p, u8p, u8p, u8p, u8p, u32p, u32p]
    L.agc_hip_fetch_slices_dev.argtypes = [vp, C.c_uint32, vp, u64p, u32p, u8p, u8p, C.c_uint64, u64p]
    L.agc_hip_ref_lag_counts_dev.argtypes = [vp, C.c_uint32, vp, u64p, u32p, u8p, u32p, u32p]
    L.agc_hip_zstd17_batch.argtypes = [vp, C.c_uint32, u8p, u64p, u8p, C.c_uint64, u64p]
    L.agc_hip_zstd17_cparams.argtypes = [C.c_uint64, u32p]
    L.agc_hip_zstd_cparams.argtypes = [C.c_int, C.c_uint64, u32p]
    L.agc_hip_zstd_batch.argtypes = [vp, C.c_uint32, u8p, u64p, u8p, u8p, C.c_uint64, u64p]
    L.agc_hip_zstd17_batch_dev.argtypes = [vp, C.c_uint32, vp, u64p, u8p, C.c_uint64, u64p]
    L.agc_hip_zstd17_background.argtypes = [vp, C.c_int]
    L.agc_hip_zstd17_max_input.restype = C.c_uint32
    L.agc_hip_zstd17_resident_frames.argtypes = [vp]
    L.agc_hip_zstd17_resident_frames.restype = C.c_uint32
    L.agc_hip_packed_words_bytes.restype = C.c_uint64
    L.agc_hip_packed_words_bytes.argtypes = [C.c_uint64]
    L.agc_hip_packed_index_bytes.restype = C.c_uint64
    L.agc_hip_packed_index_bytes.argtypes = [C.c_uint64]
    L.agc_hip_pack_dev.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, u64p]
    L.agc_hip_pack_fasta_begin.argtypes = [vp, vp, C.c_uint64, u64p, u64p, C.c_uint32, vp, vp, vp, C.c_uint64]
    L.agc_hip_pack_fasta_end.argtypes = [vp, u64p, u64p]
    L.agc_hip_pack_fasta_dev.argtypes = [vp, vp, C.c_uint64, u64p, u64p, C.c_uint32, vp, vp, vp, C.c_uint64, u64p, u64p]
    L.agc_hip_expand_dev.argtypes = [vp, C.POINTER(Packed), vp]
    L.agc_hip_scan_packed_dev.argtypes = [vp, C.POINTER(Packed), u64p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, u32p, u64p, u64p, u64p]
    L.agc_hip_prefetch_packed_dev.argtypes = [vp, C.POINTER(Packed), u64p, C.c_uint32, C.c_uint32]
    pkp = C.POINTER(Packed)
    L.agc_hip_sample_pack.argtypes = [vp, vp, C.c_uint64, pkp]
    L.agc_hip_ref_store_begin_packed.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, pkp, u64p, C.POINTER(C.c_uint32), vp, vp, vp, vp, C.c_uint64, u64p]
    L.agc_hip_ref_store_end.argtypes = [vp, C.c_uint32]
    L.agc_hip_sample_pack_fasta.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), u64p, pkp, u64p]
    L.agc_hip_ref_register_batch_packed.argtypes = [vp, C.c_uint32, u32p, pkp, u64p, u32p, u8p, C.c_uint32]
    L.agc_hip_lz_encode_batch_packed.argtypes = [vp, C.c_uint32, u32p, pkp, u64p, u32p, u8p, u8p, C.c_uint64, u64p]
    L.agc_hip_lz_encode_begin_packed.argtypes = [vp, C.c_uint32, u32p, pkp, u64p, u32p, u8p]
    L.agc_hip_lz_encode_begin_packed_on.argtypes = [vp, C.c_uint32, C.c_uint32, u32p, pkp, u64p, u32p, u8p]
    L.agc_hip_lz_encode_end_on.argtypes = [vp, C.c_uint32, u8p, C.c_uint64, u64p]
    L.agc_hip_lz_encode_pending_on.argtypes = [vp, C.c_uint32, u32p]
    L.agc_hip_lz_encode_drop_on.argtypes = [vp, C.c_uint32]
    L.agc_hip_lz_estimate_batch_packed.argtypes = [vp, C.c_uint32, u32p, pkp, u64p, u32p, u8p, u32p, u32p]
    L.agc_hip_lz_cost_vector_batch_packed.argtypes = [vp, C.c_uint32, u32p, pkp, u64p, u32p, u8p, u8p, u32p]
    L.agc_hip_lz_split_point_batch_packed.argtypes = [vp, C.c_uint32, u32p, u32p, pkp, u64p, u32p, u8p, u8p, u8p, u8p, u32p, u32p]
    L.agc_hip_fetch_slices_packed.argtypes = [vp, C.c_uint32, pkp, u64p, u32p, u8p, u8p, C.c_uint64, u64p]
    L.agc_hip_ref_lag_counts_packed.argtypes = [vp, C.c_uint32, pkp, u64p, u32p, u8p, u32p, u32p]
    L.agc_hip_group_hash.argtypes = [C.c_uint64, C.c_uint64]
    L.agc_hip_group_hash.restype = C.c_uint64
    L.agc_hip_group_map_set.argtypes = [vp, vp, C.c_uint64]
    L.agc_hip_group_map_update.argtypes = [vp, C.c_uint32, u64p, vp]
    L.agc_hip_segments_packed.argtypes = [vp, pkp, u64p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint64, vp, u64p, u32p]
    L.agc_hip_segments_encode_known.argtypes = [vp]
    L.agc_hip_scan_prefetched.argtypes = [vp, C.POINTER(Packed), u64p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, u32p, u64p, u64p, u64p]
    for s in SYMBOLS:
        f = getattr(L, s)
        if s not in ("agc_hip_destroy", "agc_hip_last_error", "agc_hip_abi_version", "agc_hip_splitters_count", "agc_hip_zstd17_max_input", "agc_hip_zstd17_resident_frames",
                     "agc_hip_packed_words_bytes", "agc_hip_packed_index_bytes", "agc_hip_group_hash"):
            f.restype = C.c_int
    _lib = L
    return L


def _a(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Context:
    """One agc_hip_ctx (one GPU, one stream)."""

    def __init__(self, device=0):
        self.L = load()
        h = vp()
        rc = self.L.agc_hip_create(C.byref(h), device)
        if rc != OK:
            raise AgcHipError(rc, "agc_hip_create failed (no HIP device?)")
        self.h = h

    @classmethod
    def from_handle(cls, handle):
        """a view of an agc_hip_ctx that somebody else owns (the host compressor's): close() leaves it alone"""
        self = cls.__new__(cls)
        self.L = load()
        self.h = vp(handle) if not isinstance(handle, vp) else handle
        self.borrowed = True
        return self

    def close(self):
        if getattr(self, "h", None) and not getattr(self, "borrowed", False):
            self.L.agc_hip_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != OK:
            raise AgcHipError(rc, self.L.agc_hip_last_error(self.h).decode())

    # ---- timing -----------------------------------------------------------
    def timing(self, on=True):
        self._chk(self.L.agc_hip_timing_enable(self.h, int(on)))
        self._chk(self.L.agc_hip_timing_reset(self.h))

    def timing_get(self):
        out = {}
        for i, n in enumerate(K_NAMES):
            ms = C.c_double()
            ln = C.c_uint64()
            self._chk(self.L.agc_hip_timing_get(self.h, i, C.byref(ms), C.byref(ln)))
            out[n] = (ms.value, ln.value)
        return out

    # ---- a1 -----------------------------------------------------------------
    def preprocess_dev(self, d_raw, n_raw, d_codes):
        n = C.c_uint64()
        self._chk(self.L.agc_hip_preprocess_dev(self.h, d_raw, n_raw, d_codes, C.byref(n)))
        return n.value

    # ---- splitters / scan ---------------------------------------------------
    def splitters_set(self, kmers):
        k = _a(kmers, np.uint64)
        self._chk(self.L.agc_hip_splitters_set(self.h, _p(k, u64p), k.size))

    def splitters_insert(self, kmers):
        k = _a(kmers, np.uint64)
        self._chk(self.L.agc_hip_splitters_insert(self.h, _p(k, u64p), k.size))

    def splitters_count(self):
        return int(self.L.agc_hip_splitters_count(self.h))

    def determine_splitters_dev(self, d_codes, ctg_off, k, segment_size, want_sorted=False):
        """-> sorted unique splitters (and, if asked, all canonical k-mers of the reference, sorted)"""
        off = _a(ctg_off, np.uint64)
        cap = 1 << 16
        total = int(off[-1] - off[0])
        srt = np.zeros(total if want_sorted else 0, np.uint64)
        while True:
            out = np.zeros(cap, np.uint64)
            n = C.c_uint64()
            ns = C.c_uint64()
            rc = self.L.agc_hip_determine_splitters_dev(self.h, d_codes, _p(off, u64p), off.size - 1, k, segment_size, cap, _p(out, u64p),
                                                        C.byref(n), srt.size, _p(srt, u64p) if want_sorted else None,
                                                        C.byref(ns) if want_sorted else None)
            if rc == ECAP and n.value > cap:
                cap = int(n.value)
                continue
            self._chk(rc)
            return (out[:n.value], srt[:ns.value]) if want_sorted else out[:n.value]

    def _scan(self, fn, codes_arg, ctg_off, k, cap):
        off = _a(ctg_off, np.uint64)
        n_ctg = off.size - 1
        while True:
            n = C.c_uint64()
            ctg = np.zeros(cap, np.uint32)
            pos = np.zeros(cap, np.uint64)
            d = np.zeros(cap, np.uint64)
            r = np.zeros(cap, np.uint64)
            rc = fn(self.h, codes_arg, _p(off, u64p), n_ctg, k, cap, C.byref(n), _p(ctg, u32p), _p(pos, u64p), _p(d, u64p), _p(r, u64p))
            if rc == ECAP:
                cap = int(n.value)
                continue
            self._chk(rc)
            m = int(n.value)
            return ctg[:m], pos[:m], d[:m], r[:m]

    def scan_contigs_dev(self, d_codes, ctg_off, k, cap=1 << 16):
        """-> (ctg, pos, dir, rc) of the accepted splitter hits."""
        return self._scan(self.L.agc_hip_scan_contigs_dev, d_codes, ctg_off, k, cap)

    # ---- 2-bit packed samples ------------------------------------------------
    def pack_dev(self, d_codes_tensor, n_symbols=None):
        """torch uint8 tensor of codes on the GPU -> (Packed struct, tensors that back it).  The escape buffer is sized by a
        first attempt and grown if the sample has more escaped blocks."""
        import torch
        n = int(d_codes_tensor.numel() if n_symbols is None else n_symbols)
        dev = d_codes_tensor.device
        torch.cuda.synchronize(dev)  # the codes come from torch's stream, the packing runs on the library's
        words = torch.empty(int(self.L.agc_hip_packed_words_bytes(n)) // 4 + 1, dtype=torch.int32, device=dev)
        index = torch.empty(int(self.L.agc_hip_packed_index_bytes(n)) // 4 + 1, dtype=torch.int32, device=dev)
        cap = 64
        while True:
            esc = torch.empty(max(cap, 1) * 1024, dtype=torch.uint8, device=dev)
            cnt = np.zeros(1, np.uint64)
            rc = self.L.agc_hip_pack_dev(self.h, d_codes_tensor.data_ptr(), n, words.data_ptr(), index.data_ptr(), esc.data_ptr(), cap, _p(cnt, u64p))
            if rc == ECAP:
                cap = int(cnt[0]) + 16
                continue
            self._chk(rc)
            break
        pk = Packed(words.data_ptr(), index.data_ptr(), esc.data_ptr(), n)
        return pk, (words, index, esc)

    def pack_fasta_begin(self, d_raw_tensor, n_raw, raw_begin, raw_end, esc_cap=64, bufs=None):
        """raw FASTA bodies in HBM (torch uint8 tensor; contig c = bytes [raw_begin[c], raw_end[c])) -> the packed sample, queued on
        the library's pack stream.  Returns the pending handle pack_fasta_end() takes.  `bufs` = (words, index, esc) of an earlier
        pack to write into (their sizes must fit)."""
        import torch
        dev = d_raw_tensor.device
        rb, re_ = _a(raw_begin, np.uint64), _a(raw_end, np.uint64)
        ub = int((re_ - rb).sum())  # kept bytes <= the ranges' lengths
        if bufs is None:
            words = torch.empty(int(self.L.agc_hip_packed_words_bytes(ub)) // 4 + 4, dtype=torch.int32, device=dev)
            index = torch.empty(int(self.L.agc_hip_packed_index_bytes(ub)) // 4 + 1, dtype=torch.int32, device=dev)
            esc = torch.empty(max(esc_cap, 1) * 1024, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize(dev)  # (torch's allocations / the raw bytes come from torch's stream)
        else:
            words, index, esc = bufs
            esc_cap = esc.numel() // 1024
        self._chk(self.L.agc_hip_pack_fasta_begin(self.h, d_raw_tensor.data_ptr(), int(n_raw), _p(rb, u64p), _p(re_, u64p), rb.size, words.data_ptr(),
                                                  index.data_ptr(), esc.data_ptr(), esc_cap))
        return {"raw": d_raw_tensor, "n_raw": int(n_raw), "rb": rb, "re": re_, "bufs": (words, index, esc)}

    def pack_fasta_end(self, pending):
        """-> (Packed, backing tensors, symbol offsets of the contigs); grows the escape buffer and packs again when it was too small"""
        import torch
        while True:
            off = np.zeros(pending["rb"].size + 1, np.uint64)
            cnt = np.zeros(1, np.uint64)
            rc = self.L.agc_hip_pack_fasta_end(self.h, _p(off, u64p), _p(cnt, u64p))
            if rc == ECAP:
                words, index, _esc = pending["bufs"]
                esc = torch.empty((int(cnt[0]) + 16) * 1024, dtype=torch.uint8, device=words.device)
                torch.cuda.synchronize(words.device)
                pending = self.pack_fasta_begin(pending["raw"], pending["n_raw"], pending["rb"], pending["re"], bufs=(words, index, esc))
                continue
            self._chk(rc)
            words, index, esc = pending["bufs"]
            return Packed(words.data_ptr(), index.data_ptr(), esc.data_ptr(), int(off[-1])), pending["bufs"], off

    def pack_fasta_dev(self, d_raw_tensor, n_raw, raw_begin, raw_end, esc_cap=64):
        return self.pack_fasta_end(self.pack_fasta_begin(d_raw_tensor, n_raw, raw_begin, raw_end, esc_cap))

    def expand_dev(self, pk, d_codes_ptr):
        self._chk(self.L.agc_hip_expand_dev(self.h, C.byref(pk), d_codes_ptr))
        self._chk(self.L.agc_hip_sync(self.h))

    def scan_packed_dev(self, pk, ctg_off, k, cap=1 << 16):
        fn = lambda h, arg, *rest: self.L.agc_hip_scan_packed_dev(h, C.byref(arg), *rest)
        return self._scan(fn, pk, ctg_off, k, cap)

    def prefetch_packed_dev(self, pk, ctg_off, k):
        """queues the scan of the NEXT sample on the prefetch stream"""
        off = _a(ctg_off, np.uint64)
        self._chk(self.L.agc_hip_prefetch_packed_dev(self.h, C.byref(pk), _p(off, u64p), off.size - 1, k))

    def sample_pack(self, d_codes_ptr, n_symbols):
        """codes in HBM -> the context's own packed buffers (valid until the next call)"""
        pk = Packed()
        self._chk(self.L.agc_hip_sample_pack(self.h, d_codes_ptr, n_symbols, C.byref(pk)))
        return pk

    def sample_pack_fasta(self, bodies):
        """FASTA bodies in host memory (a list of bytes objects, one per contig) -> (Packed in the context's own buffers, the
        contigs' symbol offsets): what AddSampleFiles does with a window of raw contigs"""
        n = len(bodies)
        ptr = (C.c_char_p * max(n, 1))(*bodies)
        ln = np.array([len(b) for b in bodies], np.uint64)
        off = np.zeros(n + 1, np.uint64)
        pk = Packed()
        self._chk(self.L.agc_hip_sample_pack_fasta(self.h, n, ptr, _p(ln, u64p), C.byref(pk), _p(off, u64p)))
        return pk, off

    def scan_prefetched(self, pk, ctg_off, k, cap=1 << 16):
        fn = lambda h, arg, *rest: self.L.agc_hip_scan_prefetched(h, C.byref(arg), *rest)
        return self._scan(fn, pk, ctg_off, k, cap)

    # ---- segments and their groups on the device ------------------------------
    def group_map_set(self, keys_to_gid, n_slots=None):
        """{(k1, k2): gid} -> the open-addressing array of include/agc_hip.h, uploaded whole; returns the numpy table"""
        n = max(16, n_slots or 16)
        while n < 2 * len(keys_to_gid):
            n *= 2
        tab = np.zeros(n, GROUP_SLOT_DTYPE)
        for (k1, k2), gid in keys_to_gid.items():
            i = int(self.L.agc_hip_group_hash(k1, k2)) & (n - 1)
            while tab[i]["used"]:
                i = (i + 1) & (n - 1)
            tab[i] = (k1, k2, gid, 1)
        self._chk(self.L.agc_hip_group_map_set(self.h, tab.ctypes.data, n))
        return tab

    def group_map_update(self, idx, slots):
        idx = _a(idx, np.uint64)
        slots = np.ascontiguousarray(slots, dtype=GROUP_SLOT_DTYPE)
        self._chk(self.L.agc_hip_group_map_update(self.h, idx.size, _p(idx, u64p), slots.ctypes.data))

    def segments_packed(self, pk, ctg_off, k, prefetched=False, encode_known=False, cap=1 << 12):
        """-> (segments as a structured numpy array, number of deltas the launched encode will deliver)"""
        off = _a(ctg_off, np.uint64)
        while True:
            segs = np.zeros(cap, SEGMENT_DTYPE)
            n = C.c_uint64()
            ne = C.c_uint32()
            rc = self.L.agc_hip_segments_packed(self.h, C.byref(pk), _p(off, u64p), off.size - 1, k, int(prefetched), int(encode_known), cap, segs.ctypes.data,
                                                C.byref(n), C.byref(ne))
            if rc == ECAP:
                cap = int(n.value)
                continue
            self._chk(rc)
            if encode_known:
                ln = segs["len"][:n.value][segs["encoded"][:n.value] != 0].astype(np.uint32)
                self._enc_pending = (np.zeros(ln.size, np.uint32), None, ln, None)  # (what lz_encode_end sizes its buffers by)
            return segs[:n.value], int(ne.value)

    def segments_encode_known(self, segs, known_gids):
        """launches the encode of the segments of the last segments_packed call whose group the table knew (second lane);
        known_gids: the groups with a registered reference -- what tells the caller which deltas lz_encode_end will deliver"""
        self._chk(self.L.agc_hip_segments_encode_known(self.h))
        m = (segs["front_full"] != 0) & (segs["back_full"] != 0) & (segs["map_gid"] >= 16) & np.isin(segs["map_gid"], np.asarray(list(known_gids), np.int64))
        ln = segs["len"][m].astype(np.uint32)
        self._enc_pending = (np.zeros(ln.size, np.uint32), None, ln, None)
        return m

    def scan_contigs(self, codes, ctg_off, k, cap=1 << 16):
        codes = _a(codes, np.uint8)
        return self._scan(self.L.agc_hip_scan_contigs, _p(codes, u8p), ctg_off, k, cap)

    # ---- references ---------------------------------------------------------
    def ref_register(self, gid, ref, min_match_len):
        ref = _a(ref, np.uint8)
        self._chk(self.L.agc_hip_ref_register(self.h, gid, _p(ref, u8p), ref.size, min_match_len))

    def ref_register_batch_dev(self, gids, d_base, off, length, rc, min_match_len):
        g, o, l = _a(gids, np.uint32), _a(off, np.uint64), _a(length, np.uint32)
        r = _a(rc, np.uint8) if rc is not None else None
        self._chk(self.L.agc_hip_ref_register_batch_dev(self.h, g.size, _p(g, u32p), d_base, _p(o, u64p), _p(l, u32p), _p(r, u8p), min_match_len))

    def ref_register_batch_packed(self, gids, pk, off, length, rc, min_match_len):
        g, o, l = _a(gids, np.uint32), _a(off, np.uint64), _a(length, np.uint32)
        r = _a(rc, np.uint8) if rc is not None else None
        self._chk(self.L.agc_hip_ref_register_batch_packed(self.h, g.size, _p(g, u32p), C.byref(pk), _p(o, u64p), _p(l, u32p), _p(r, u8p), min_match_len))

    def ref_get(self, gid):
        n = C.c_uint32()
        rc = self.L.agc_hip_ref_get(self.h, gid, None, 0, C.byref(n))
        if rc not in (OK, ECAP):
            self._chk(rc)
        out = np.zeros(n.value, np.uint8)
        self._chk(self.L.agc_hip_ref_get(self.h, gid, _p(out, u8p), out.size, C.byref(n)))
        return out

    def ref_index_get(self, gid):
        hs = C.c_uint64()
        is16 = C.c_int()
        rc = self.L.agc_hip_ref_index_get(self.h, gid, None, 0, C.byref(hs), C.byref(is16))
        if rc not in (OK, ECAP):
            self._chk(rc)
        out = np.zeros(hs.value, np.uint32)
        self._chk(self.L.agc_hip_ref_index_get(self.h, gid, _p(out, u32p), out.size, C.byref(hs), C.byref(is16)))
        return out, bool(is16.value)

    # ---- parse batches ------------------------------------------------------
    @staticmethod
    def _batch(gids, off, length, rc):
        g, o, l = _a(gids, np.uint32), _a(off, np.uint64), _a(length, np.uint32)
        r = _a(rc, np.uint8) if rc is not None else None
        assert g.size == o.size == l.size and (r is None or r.size == g.size)
        return g, o, l, r

    def _encode(self, fn, base, gids, off, length, rc, enc_cap):
        g, o, l, r = self._batch(gids, off, length, rc)
        if enc_cap is None:
            enc_cap = int(l.astype(np.uint64).sum()) * 21 // 16 + 64 * g.size + 64
        enc = np.empty(enc_cap, np.uint8)
        eoff = np.zeros(g.size + 1, np.uint64)
        self._chk(fn(self.h, g.size, _p(g, u32p), base, _p(o, u64p), _p(l, u32p), _p(r, u8p), _p(enc, u8p), enc_cap, _p(eoff, u64p)))
        return enc[:int(eoff[-1])], eoff

    def lz_encode_batch_dev(self, d_base, gids, off, length, rc=None, enc_cap=None):
        """-> (enc bytes, enc_off[n+1])"""
        return self._encode(self.L.agc_hip_lz_encode_batch_dev, d_base, gids, off, length, rc, enc_cap)

    def lz_encode_begin_dev(self, d_base, gids, off, length, rc=None):
        """first half of lz_encode_batch_dev: queues the encode on the context's second stream and returns"""
        g, o, l, r = self._batch(gids, off, length, rc)
        self._enc_pending = (g, o, l, r)  # (the arrays stay alive until the end call)
        self._chk(self.L.agc_hip_lz_encode_begin_dev(self.h, g.size, _p(g, u32p), d_base, _p(o, u64p), _p(l, u32p), _p(r, u8p)))

    # the same entry points on sequences of a packed sample (pk: Packed)
    def lz_encode_batch_packed(self, pk, gids, off, length, rc=None, enc_cap=None):
        return self._encode(self.L.agc_hip_lz_encode_batch_packed, C.byref(pk), gids, off, length, rc, enc_cap)

    def lz_encode_begin_packed(self, pk, gids, off, length, rc=None, lane=None):
        """first half of an encode on lane `lane` (None: the entry point without a lane, which is lane 0)"""
        g, o, l, r = self._batch(gids, off, length, rc)
        if lane is None:
            self._enc_pending = (g, o, l, r)
            self._chk(self.L.agc_hip_lz_encode_begin_packed(self.h, g.size, _p(g, u32p), C.byref(pk), _p(o, u64p), _p(l, u32p), _p(r, u8p)))
        else:
            if not hasattr(self, "_enc_pending_on"):
                self._enc_pending_on = {}
            self._enc_pending_on[lane] = (g, o, l, r)
            self._chk(self.L.agc_hip_lz_encode_begin_packed_on(self.h, lane, g.size, _p(g, u32p), C.byref(pk), _p(o, u64p), _p(l, u32p), _p(r, u8p)))

    def lz_encode_end_on(self, lane, enc_cap=None):
        """second half on a lane: -> (enc bytes, enc_off[n+1])"""
        g, o, l, r = self._enc_pending_on[lane]
        n_dev = C.c_uint32()
        self._chk(self.L.agc_hip_lz_encode_pending_on(self.h, lane, C.byref(n_dev)))
        assert n_dev.value == g.size, (n_dev.value, g.size)
        if enc_cap is None:
            enc_cap = int(l.astype(np.uint64).sum()) * 21 // 16 + 64 * g.size + 64
        enc = np.empty(enc_cap, np.uint8)
        eoff = np.zeros(g.size + 1, np.uint64)
        rc_ = self.L.agc_hip_lz_encode_end_on(self.h, lane, _p(enc, u8p), enc_cap, _p(eoff, u64p))
        if rc_ == ECAP:
            enc_cap = int(eoff[-1]) + 64
            enc = np.empty(enc_cap, np.uint8)
            rc_ = self.L.agc_hip_lz_encode_end_on(self.h, lane, _p(enc, u8p), enc_cap, _p(eoff, u64p))
        self._chk(rc_)
        del self._enc_pending_on[lane]
        return enc[:int(eoff[-1])], eoff

    def lz_estimate_batch_packed(self, pk, gids, off, length, rc=None):
        return self._estimate(self.L.agc_hip_lz_estimate_batch_packed, C.byref(pk), gids, off, length, rc)

    def lz_cost_vector_batch_packed(self, pk, gids, off, length, rc, prefix):
        return self._costvec(self.L.agc_hip_lz_cost_vector_batch_packed, C.byref(pk), gids, off, length, rc, prefix)

    def lz_split_point_batch_packed(self, pk, gid1, gid2, off, length, rc1, prefix1, rc2, prefix2):
        return self._split(self.L.agc_hip_lz_split_point_batch_packed, C.byref(pk), gid1, gid2, off, length, rc1, prefix1, rc2, prefix2)

    def fetch_slices_packed(self, pk, off, length, rc=None):
        return self._fetch(self.L.agc_hip_fetch_slices_packed, C.byref(pk), off, length, rc)

    def ref_lag_counts_packed(self, pk, off, length, rc=None):
        return self._lag(self.L.agc_hip_ref_lag_counts_packed, C.byref(pk), off, length, rc)

    def lz_encode_end(self, enc_cap=None):
        """second half: waits, -> (enc bytes, enc_off[n+1]) exactly as lz_encode_batch_dev"""
        g, o, l, r = self._enc_pending
        n_dev = C.c_uint32()
        self._chk(self.L.agc_hip_lz_encode_pending(self.h, C.byref(n_dev)))
        assert n_dev.value == g.size, (n_dev.value, g.size)
        if enc_cap is None:
            enc_cap = int(l.astype(np.uint64).sum()) * 21 // 16 + 64 * g.size + 64
        enc = np.empty(enc_cap, np.uint8)
        eoff = np.zeros(g.size + 1, np.uint64)
        rc_ = self.L.agc_hip_lz_encode_end(self.h, _p(enc, u8p), enc_cap, _p(eoff, u64p))
        if rc_ == ECAP:
            enc_cap = int(eoff[-1]) + 64
            enc = np.empty(enc_cap, np.uint8)
            rc_ = self.L.agc_hip_lz_encode_end(self.h, _p(enc, u8p), enc_cap, _p(eoff, u64p))
        self._chk(rc_)
        self._enc_pending = None
        return enc[:int(eoff[-1])], eoff

    def lz_encode_batch(self, text, gids, off, length, rc=None, enc_cap=None):
        text = _a(text, np.uint8)
        return self._encode(self.L.agc_hip_lz_encode_batch, _p(text, u8p), gids, off, length, rc, enc_cap)

    def _estimate(self, fn, base, gids, off, length, rc):
        g, o, l, r = self._batch(gids, off, length, rc)
        cost = np.zeros(g.size, np.uint32)
        peak = np.zeros(g.size, np.uint32)
        self._chk(fn(self.h, g.size, _p(g, u32p), base, _p(o, u64p), _p(l, u32p), _p(r, u8p), _p(cost, u32p), _p(peak, u32p)))
        return cost, peak

    def lz_estimate_batch_dev(self, d_base, gids, off, length, rc=None):
        return self._estimate(self.L.agc_hip_lz_estimate_batch_dev, d_base, gids, off, length, rc)

    def lz_estimate_batch(self, text, gids, off, length, rc=None):
        text = _a(text, np.uint8)
        return self._estimate(self.L.agc_hip_lz_estimate_batch, _p(text, u8p), gids, off, length, rc)

    def _costvec(self, fn, base, gids, off, length, rc, prefix):
        g, o, l, r = self._batch(gids, off, length, rc)
        pf = _a(prefix, np.uint8)
        costs = np.zeros(int(l.astype(np.uint64).sum()), np.uint32)
        self._chk(fn(self.h, g.size, _p(g, u32p), base, _p(o, u64p), _p(l, u32p), _p(r, u8p), _p(pf, u8p), _p(costs, u32p)))
        return costs

    def lz_cost_vector_batch_dev(self, d_base, gids, off, length, rc, prefix):
        return self._costvec(self.L.agc_hip_lz_cost_vector_batch_dev, d_base, gids, off, length, rc, prefix)

    def lz_cost_vector_batch(self, text, gids, off, length, rc, prefix):
        text = _a(text, np.uint8)
        return self._costvec(self.L.agc_hip_lz_cost_vector_batch, _p(text, u8p), gids, off, length, rc, prefix)

    def _split(self, fn, base, gid1, gid2, off, length, rc1, prefix1, rc2, prefix2):
        g1, g2, o, l = _a(gid1, np.uint32), _a(gid2, np.uint32), _a(off, np.uint64), _a(length, np.uint32)
        r1, p1, r2, p2 = (_a(x, np.uint8) for x in (rc1, prefix1, rc2, prefix2))
        pos = np.zeros(g1.size, np.uint32)
        sm = np.zeros(g1.size, np.uint32)
        self._chk(fn(self.h, g1.size, _p(g1, u32p), _p(g2, u32p), base, _p(o, u64p), _p(l, u32p),
                     _p(r1, u8p), _p(p1, u8p), _p(r2, u8p), _p(p2, u8p), _p(pos, u32p), _p(sm, u32p)))
        return pos, sm

    def lz_split_point_batch_dev(self, d_base, gid1, gid2, off, length, rc1, prefix1, rc2, prefix2):
        return self._split(self.L.agc_hip_lz_split_point_batch_dev, d_base, gid1, gid2, off, length, rc1, prefix1, rc2, prefix2)

    def zstd17_batch(self, inputs):
        """inputs: list of bytes-like; returns the list of level-17 zstd frames (S3 on the GPU)"""
        n = len(inputs)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(x) for x in inputs])
        src = np.frombuffer(b"".join(bytes(x) for x in inputs), np.uint8) if off[-1] else np.zeros(1, np.uint8)
        cap = int(off[-1]) + 32 * n + 64
        dst = np.zeros(cap, np.uint8)
        doff = np.zeros(n + 1, np.uint64)
        self._chk(self.L.agc_hip_zstd17_batch(self.h, n, _p(src, u8p), _p(off, u64p), _p(dst, u8p), cap, _p(doff, u64p)))
        return [dst[int(doff[i]):int(doff[i + 1])].tobytes() for i in range(n)]

    def zstd_batch(self, inputs, levels):
        """inputs: list of bytes-like, levels: 13 / 17 / 19 per input; returns the list of zstd frames (S3 on the GPU)"""
        n = len(inputs)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum([len(x) for x in inputs])
        src = np.frombuffer(b"".join(bytes(x) for x in inputs), np.uint8) if off[-1] else np.zeros(1, np.uint8)
        lv = np.ascontiguousarray(levels, dtype=np.uint8)
        assert lv.size == n
        cap = int(off[-1]) + 32 * n + 64
        dst = np.zeros(cap, np.uint8)
        doff = np.zeros(n + 1, np.uint64)
        self._chk(self.L.agc_hip_zstd_batch(self.h, n, _p(src, u8p), _p(off, u64p), _p(lv, u8p), _p(dst, u8p), cap, _p(doff, u64p)))
        return [dst[int(doff[i]):int(doff[i + 1])].tobytes() for i in range(n)]

    def zstd17_batch_raw(self, src, off):
        """packs back to back (pack i = src[off[i]:off[i+1]]) -> (frames back to back, their offsets [n + 1]); no per-pack
        Python work (the multi-GPU Close hands tens of thousands of packs through here)"""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        cap = int(off[-1] - off[0]) + 32 * n + 64
        dst = np.empty(cap, np.uint8)
        doff = np.zeros(n + 1, np.uint64)
        if src.size == 0:
            src = np.zeros(1, np.uint8)
        self._chk(self.L.agc_hip_zstd17_batch(self.h, n, _p(src, u8p), _p(off, u64p), _p(dst, u8p), cap, _p(doff, u64p)))
        return dst[:int(doff[-1])], doff

    def zstd17_batch_raw_dev(self, d_src_ptr, off):
        """the same with the packs already in HBM (d_src_ptr: device address of pack 0's first byte minus off[0])"""
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = off.size - 1
        cap = int(off[-1] - off[0]) + 32 * n + 64
        dst = np.empty(cap, np.uint8)
        doff = np.zeros(n + 1, np.uint64)
        self._chk(self.L.agc_hip_zstd17_batch_dev(self.h, n, d_src_ptr, _p(off, u64p), _p(dst, u8p), cap, _p(doff, u64p)))
        return dst[:int(doff[-1])], doff

    def _fetch(self, fn, base, off, length, rc):
        o, l = _a(off, np.uint64), _a(length, np.uint32)
        r = _a(rc, np.uint8) if rc is not None else None
        cap = int(l.astype(np.uint64).sum())
        out = np.empty(cap, np.uint8)
        ooff = np.zeros(o.size + 1, np.uint64)
        self._chk(fn(self.h, o.size, base, _p(o, u64p), _p(l, u32p), _p(r, u8p), _p(out, u8p), cap, _p(ooff, u64p)))
        return out, ooff

    def fetch_slices_dev(self, d_base, off, length, rc=None):
        return self._fetch(self.L.agc_hip_fetch_slices_dev, d_base, off, length, rc)

    def _lag(self, fn, base, off, length, rc):
        o, l = _a(off, np.uint64), _a(length, np.uint32)
        r = _a(rc, np.uint8) if rc is not None else None
        cnt = np.zeros((o.size, 28), np.uint32)
        cur = np.zeros((o.size, 28), np.uint32)
        self._chk(fn(self.h, o.size, base, _p(o, u64p), _p(l, u32p), _p(r, u8p), _p(cnt, u32p), _p(cur, u32p)))
        return cnt, cur

    def ref_lag_counts_dev(self, d_base, off, length, rc=None):
        return self._lag(self.L.agc_hip_ref_lag_counts_dev, d_base, off, length, rc)
