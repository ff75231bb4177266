"""ctypes binding of the host-side compressor (libagc_host.so, agc_amd/csrc/host/): the
reference's CAGCCompressor::Create / AddSampleFiles / Close interface for the create path
(src/core/agc_compressor.h:754-763) plus AddSampleDevice for HBM-resident samples."""
import ctypes as C
import os

import numpy as np

from . import capi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libagc_host.so")
STAT_NAMES = ["bases", "segments", "new_groups", "one_splitter", "middle_tried", "middle_split", "lz_encoded", "delta_bytes",
              "ref_bytes", "zstd_in", "zstd_out", "archive_bytes",
              "t_scan", "t_classify", "t_gpu_aux", "t_register", "t_encode", "t_store", "t_zstd", "t_io", "t_device",
              "h_scan", "h_classify", "h_gpu_aux", "h_register", "h_encode", "h_store", "windows", "commit_runs", "revalidated",
              "enc_text", "enc_ref", "est_text", "est_ref", "cv_text", "cv_ref", "zstd_dev_in", "t_zstd_dev", "t_zstd_host", "t_zstd_stage", "t_zstd_wait", "reprepared", "zstd_dev_out", "windows_cut"]

_lib = None


def bind(L):
    """declares the argument types of the agc_cmp_* entry points (agc_amd/csrc/host/capi_host.cpp) on a loaded library"""
    vp = C.c_void_p
    L.agc_cmp_new.restype = vp
    L.agc_cmp_new.argtypes = [C.c_int]
    L.agc_cmp_delete.argtypes = [vp]
    L.agc_cmp_delete.restype = None
    L.agc_cmp_create.argtypes = [vp, C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                 C.c_uint32, C.c_uint32, C.c_double]
    L.agc_cmp_set_splitters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_uint64]
    L.agc_cmp_set_reference_dev.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.c_uint32]
    L.agc_cmp_add_sample_files.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32]
    L.agc_cmp_add_sample_dev.argtypes = [vp, C.c_char_p, C.c_uint32, C.POINTER(C.c_char_p), vp, C.POINTER(C.c_uint64)]
    L.agc_cmp_close.argtypes = [vp, C.c_uint32]
    L.agc_cmp_drain.argtypes = [vp]
    L.agc_cmp_close_collect_packs.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint32)]
    L.agc_cmp_close_provide_frames.argtypes = [vp, vp, vp]
    L.agc_cmp_deferred_pack_bytes.argtypes = [vp]
    L.agc_cmp_deferred_pack_bytes.restype = C.c_uint64
    L.agc_cmp_deal_collect_packs.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_uint32)]
    L.agc_cmp_deal_keep_own.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.agc_cmp_deal_provide_frames.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.agc_cmp_zstd_version.argtypes = [vp]
    L.agc_cmp_zstd_version.restype = C.c_char_p
    L.agc_cmp_hip_ctx.argtypes = [vp]
    L.agc_cmp_hip_ctx.restype = vp
    L.agc_cmp_stats.argtypes = [vp, C.POINTER(C.c_double), C.c_uint32]
    L.agc_cmp_set_distributed.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.agc_cmp_last_record.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
    L.agc_cmp_last_record_framed.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
    L.agc_cmp_apply_record.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64]
    L.agc_cmp_last_record_body.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]
    L.agc_cmp_record_body_buffer.argtypes = [vp, C.c_uint64, C.POINTER(C.POINTER(C.c_uint8))]
    L.agc_cmp_prepare_sample_dev.argtypes = [vp, C.c_char_p, C.c_uint32, C.POINTER(C.c_char_p), vp, C.POINTER(C.c_uint64)]
    L.agc_cmp_prepare_sample_packed_dev.argtypes = [vp, C.c_char_p, C.c_uint32, C.POINTER(C.c_char_p), vp, C.POINTER(C.c_uint64)]
    L.agc_cmp_add_sample_packed_dev.argtypes = [vp, C.c_char_p, C.c_uint32, C.POINTER(C.c_char_p), vp, C.POINTER(C.c_uint64)]
    L.agc_cmp_set_next_sample_packed_dev.argtypes = [vp, vp, C.POINTER(C.c_uint64), C.c_uint32]
    L.agc_cmp_set_next_fasta_dev.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32, vp, vp, vp, C.c_uint64]
    L.agc_cmp_finish_fasta_dev.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.agc_cmp_commit_prepared.argtypes = [vp]
    L.agc_cmp_commit_prepared_head.argtypes = [vp]
    L.agc_cmp_commit_prepared_finish.argtypes = [vp]
    L.agc_cmp_append.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32]
    return L


def load():
    global _lib
    if _lib is not None:
        return _lib
    capi.load()  # torch first, then libagc_hip.so (one HIP runtime per process)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m agc_amd.build`")
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


class Compressor:
    def __init__(self, device=0, lib=None):
        """lib: an already bound library (bind()); default = the in-tree libagc_host.so on top of libagc_hip.so"""
        self.L = lib if lib is not None else load()
        self.h = self.L.agc_cmp_new(device)

    def close_handle(self):
        if getattr(self, "h", None):
            self.L.agc_cmp_delete(self.h)
            self.h = None

    def __del__(self):
        self.close_handle()

    def create(self, out_path, pack_cardinality=50, k=31, ref_file=None, segment_size=60000, min_match_len=20,
               concatenated=False, adaptive=False, verbosity=0, n_threads=8, fallback_frac=0.0):
        ok = self.L.agc_cmp_create(self.h, (out_path or "").encode(), pack_cardinality, k, (ref_file or "").encode(), segment_size,
                                   min_match_len, int(concatenated), int(adaptive), verbosity, n_threads, fallback_frac)
        if not ok:
            raise RuntimeError("CAGCCompressor::Create failed (see stderr)")

    def set_splitters(self, kmers):
        k = np.ascontiguousarray(kmers, dtype=np.uint64)
        if not self.L.agc_cmp_set_splitters(self.h, k.ctypes.data_as(C.POINTER(C.c_uint64)), k.size):
            raise RuntimeError("SetSplitters failed")

    def set_reference_dev(self, d_codes, ctg_off):
        """determine_splitters on the GPU for a reference genome resident in HBM"""
        off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
        if not self.L.agc_cmp_set_reference_dev(self.h, d_codes, off.ctypes.data_as(C.POINTER(C.c_uint64)), off.size - 1):
            raise RuntimeError("SetReferenceDevice failed (see stderr)")

    def add_sample_files(self, pairs, n_threads=8):
        n = len(pairs)
        names = (C.c_char_p * n)(*[p[0].encode() for p in pairs])
        paths = (C.c_char_p * n)(*[p[1].encode() for p in pairs])
        if not self.L.agc_cmp_add_sample_files(self.h, n, names, paths, n_threads):
            raise RuntimeError("AddSampleFiles failed (see stderr)")

    def add_sample_dev(self, sample_name, contig_names, d_codes, ctg_off):
        n = len(contig_names)
        names = (C.c_char_p * n)(*[c.encode() for c in contig_names])
        off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
        if not self.L.agc_cmp_add_sample_dev(self.h, sample_name.encode(), n, names, d_codes, off.ctypes.data_as(C.POINTER(C.c_uint64))):
            raise RuntimeError("AddSampleDevice failed (see stderr)")

    def add_sample_packed_dev(self, sample_name, contig_names, packed, ctg_off):
        """packed: agc_amd.capi.Packed (the sample in the 2-bit HBM layout); ctg_off: symbol offsets in that buffer"""
        n = len(contig_names)
        names = (C.c_char_p * n)(*[c.encode() for c in contig_names])
        off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
        if not self.L.agc_cmp_add_sample_packed_dev(self.h, sample_name.encode(), n, names, C.byref(packed), off.ctypes.data_as(C.POINTER(C.c_uint64))):
            raise RuntimeError("AddSamplePackedDevice failed (see stderr)")

    def set_next_sample_packed_dev(self, packed, ctg_off):
        """the packed sample that will be added after the next add call: its expansion + scan run ahead on the device"""
        off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
        return bool(self.L.agc_cmp_set_next_sample_packed_dev(self.h, C.byref(packed), off.ctypes.data_as(C.POINTER(C.c_uint64)), off.size - 1))

    def set_next_fasta_dev(self, d_raw_tensor, n_raw, raw_begin, raw_end, bufs):
        """announces a sample that is still the bytes of its FASTA file in HBM (torch uint8 tensor; contig c = bytes
        [raw_begin[c], raw_end[c])); bufs = (words, index, esc) tensors the packed sample is written into.  The compressor queues
        the conversion where the sample call in progress leaves the GPU room for it.  -> the pending handle finish_fasta_dev takes"""
        rb = np.ascontiguousarray(raw_begin, dtype=np.uint64)
        re_ = np.ascontiguousarray(raw_end, dtype=np.uint64)
        words, index, esc = bufs
        u64p = C.POINTER(C.c_uint64)
        if not self.L.agc_cmp_set_next_fasta_dev(self.h, d_raw_tensor.data_ptr(), int(n_raw), rb.ctypes.data_as(u64p), re_.ctypes.data_as(u64p), rb.size,
                                                 words.data_ptr(), index.data_ptr(), esc.data_ptr(), esc.numel() // 1024):
            raise RuntimeError("SetNextFastaDevice refused (a conversion is pending)")
        return {"raw": d_raw_tensor, "n_raw": int(n_raw), "rb": rb, "re": re_, "bufs": bufs}

    def finish_fasta_dev(self, pending):
        """-> (Packed, backing tensors, the contigs' symbol offsets); grows the escape buffer and converts again when it was too small"""
        import torch
        from .capi import ECAP, OK, Packed
        while True:
            off = np.zeros(pending["rb"].size + 1, np.uint64)
            cnt = np.zeros(1, np.uint64)
            u64p = C.POINTER(C.c_uint64)
            rc = self.L.agc_cmp_finish_fasta_dev(self.h, off.ctypes.data_as(u64p), cnt.ctypes.data_as(u64p))
            words, index, esc = pending["bufs"]
            if rc == ECAP:
                esc = torch.empty((int(cnt[0]) + 16) * 1024, dtype=torch.uint8, device=words.device)
                torch.cuda.synchronize(words.device)
                pending = self.set_next_fasta_dev(pending["raw"], pending["n_raw"], pending["rb"], pending["re"], (words, index, esc))
                continue
            if rc != OK:
                raise RuntimeError(f"FinishFastaDevice failed ({rc})")
            return Packed(words.data_ptr(), index.data_ptr(), esc.data_ptr(), int(off[-1])), pending["bufs"], off

    def prepare_sample_packed_dev(self, sample_name, contig_names, packed, ctg_off):
        n = len(contig_names)
        names = (C.c_char_p * n)(*[c.encode() for c in contig_names])
        off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
        if not self.L.agc_cmp_prepare_sample_packed_dev(self.h, sample_name.encode(), n, names, C.byref(packed), off.ctypes.data_as(C.POINTER(C.c_uint64))):
            raise RuntimeError("PrepareSamplePackedDevice failed (see stderr)")

    def append(self, in_archive, out_path, concatenated=False, adaptive=False, verbosity=0, n_threads=8):
        if not self.L.agc_cmp_append(self.h, in_archive.encode(), (out_path or "").encode(), verbosity, int(concatenated), int(adaptive), n_threads):
            raise RuntimeError("CAGCCompressor::Append failed (see stderr)")

    # ---- multi-GPU single-archive mode (agc_amd/dist.py drives these) ----
    def set_distributed(self, rank, world_size, writer_rank=0):
        if not self.L.agc_cmp_set_distributed(self.h, rank, world_size, writer_rank):
            raise RuntimeError("SetDistributed failed (must precede create)")

    def prepare_sample_dev(self, sample_name, contig_names, d_codes, ctg_off):
        """scan + classification + speculative encode against the current state; d_codes must stay valid until commit_prepared()"""
        n = len(contig_names)
        names = (C.c_char_p * n)(*[c.encode() for c in contig_names])
        off = np.ascontiguousarray(ctg_off, dtype=np.uint64)
        if not self.L.agc_cmp_prepare_sample_dev(self.h, sample_name.encode(), n, names, d_codes, off.ctypes.data_as(C.POINTER(C.c_uint64))):
            raise RuntimeError("PrepareSampleDevice failed (see stderr)")

    def commit_prepared(self):
        if not self.L.agc_cmp_commit_prepared(self.h):
            raise RuntimeError("CommitPrepared failed (see stderr)")

    def commit_prepared_head(self):
        """first step of commit_prepared(): the sample is registered and last_record() (the head of its commit record) is ready"""
        if not self.L.agc_cmp_commit_prepared_head(self.h):
            raise RuntimeError("CommitPreparedHead failed (see stderr)")

    def commit_prepared_finish(self):
        """second step: new references indexed on this GPU, remaining deltas encoded, last_record_body() built, bookkeeping"""
        if not self.L.agc_cmp_commit_prepared_finish(self.h):
            raise RuntimeError("CommitPreparedFinish failed (see stderr)")

    def last_record(self, copy=True):
        """the commit record of the sample just added (numpy uint8; copy=False: a view into the compressor, valid until this
        rank's next commit -- agc_amd.dist hands it straight to the collective)"""
        p = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        self.L.agc_cmp_last_record(self.h, C.byref(p), C.byref(n))
        if not n.value:
            return np.zeros(0, np.uint8)
        a = np.ctypeslib.as_array(p, shape=(n.value,))
        return a.copy() if copy else a

    def last_record_framed(self):
        """the same record as a view into the compressor's pinned buffer WITH the 64 bytes in front of it that belong to the transport
        (agc_amd.dist writes its message header there); empty when there is no record"""
        p = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        self.L.agc_cmp_last_record_framed(self.h, C.byref(p), C.byref(n))
        if not n.value:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def last_record_body(self, copy=True):
        """the LZ deltas of the sample just added (the part of its commit record only the writer rank needs; pinned host memory)"""
        p = C.POINTER(C.c_uint8)()
        n = C.c_uint64()
        self.L.agc_cmp_last_record_body(self.h, C.byref(p), C.byref(n))
        if not n.value:
            return np.zeros(0, np.uint8)
        a = np.ctypeslib.as_array(p, shape=(n.value,))
        return a.copy() if copy else a

    def record_body_buffer(self, n):
        """writer rank: pinned host buffer (numpy uint8 view, n bytes) to receive the next record's body into; apply_record takes
        it over without a copy when it is given this buffer's address"""
        p = C.POINTER(C.c_uint8)()
        if not self.L.agc_cmp_record_body_buffer(self.h, n, C.byref(p)):
            raise RuntimeError("RecordBodyBuffer failed")
        return np.ctypeslib.as_array(p, shape=(n,))

    def apply_record(self, h_ptr, n, d_ptr=None, body_ptr=None, body_n=0):
        if not self.L.agc_cmp_apply_record(self.h, h_ptr, n, d_ptr, body_ptr, body_n):
            raise RuntimeError("ApplyRecord failed (see stderr)")

    def drain(self):
        """waits for the entropy stage that runs beside the add calls (never needed for correctness: close() drains)"""
        if not self.L.agc_cmp_drain(self.h):
            raise RuntimeError("Drain failed")

    def close_collect_packs(self):
        """(src bytes as a numpy view, offsets [n + 1]) of the pending packs a device may compress; valid until close()"""
        p = C.POINTER(C.c_uint8)()
        o = C.POINTER(C.c_uint64)()
        n = C.c_uint32()
        if not self.L.agc_cmp_close_collect_packs(self.h, C.byref(p), C.byref(o), C.byref(n)):
            raise RuntimeError("CloseCollectPacks failed")
        off = np.ctypeslib.as_array(o, shape=(n.value + 1,)).copy()
        src = np.ctypeslib.as_array(p, shape=(int(off[-1]),)) if off[-1] else np.zeros(0, np.uint8)
        return src, off

    # ---- full packs dealt to the ranks in the middle of a run (agc_amd/dist.py) ----
    def deferred_pack_bytes(self):
        return int(self.L.agc_cmp_deferred_pack_bytes(self.h))

    def deal_collect_packs(self):
        """-> (deal id, src bytes as a numpy view, offsets [n + 1]); the view stays valid until every pack of the deal is settled"""
        d = C.c_uint32()
        p = C.POINTER(C.c_uint8)()
        o = C.POINTER(C.c_uint64)()
        n = C.c_uint32()
        if not self.L.agc_cmp_deal_collect_packs(self.h, C.byref(d), C.byref(p), C.byref(o), C.byref(n)):
            raise RuntimeError("DealCollectPacks failed")
        if not n.value:
            return 0, np.zeros(0, np.uint8), np.zeros(1, np.uint64)
        off = np.ctypeslib.as_array(o, shape=(n.value + 1,)).copy()
        src = np.ctypeslib.as_array(p, shape=(int(off[-1]),)) if off[-1] else np.zeros(0, np.uint8)
        return int(d.value), src, off

    def deal_keep_own(self, deal, first, count):
        if not self.L.agc_cmp_deal_keep_own(self.h, deal, first, count):
            raise RuntimeError("DealKeepOwn failed")

    def deal_provide_frames(self, deal, first, count, frames, off):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        if not self.L.agc_cmp_deal_provide_frames(self.h, deal, first, count, frames.ctypes.data, off.ctypes.data):
            raise RuntimeError("DealProvideFrames failed")

    def close_provide_frames(self, frames, off):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        if not self.L.agc_cmp_close_provide_frames(self.h, frames.ctypes.data, off.ctypes.data):
            raise RuntimeError("CloseProvideFrames failed")

    def close(self, n_threads=8):
        if not self.L.agc_cmp_close(self.h, n_threads):
            raise RuntimeError("Close failed")

    def zstd_version(self):
        return self.L.agc_cmp_zstd_version(self.h).decode()

    def stats(self):
        v = (C.c_double * len(STAT_NAMES))()
        self.L.agc_cmp_stats(self.h, v, len(STAT_NAMES))
        return {n: v[i] for i, n in enumerate(STAT_NAMES)}

    def hip_ctx(self):
        """the compressor's agc_hip_ctx (raw handle; agc_amd.capi.Context.from_handle wraps it)"""
        return self.L.agc_cmp_hip_ctx(self.h)

    def hip_timing(self, on=True):
        L = capi.load()
        ctx = self.L.agc_cmp_hip_ctx(self.h)
        L.agc_hip_timing_enable(ctx, int(on))
        L.agc_hip_timing_reset(ctx)

    def hip_timing_get(self):
        L = capi.load()
        ctx = self.L.agc_cmp_hip_ctx(self.h)
        out = {}
        for i, n in enumerate(capi.K_NAMES):
            ms = C.c_double()
            ln = C.c_uint64()
            L.agc_hip_timing_get(ctx, i, C.byref(ms), C.byref(ln))
            out[n] = (ms.value, ln.value)
        return out
