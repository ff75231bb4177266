"""FASTA -> symbol codes on the host, the way the compressor's own reader does it (genome_io.cpp:208-252 framing,
preprocess_raw_contig agc_compressor.cpp:907-951): used by the multi-GPU front end."""
import gzip

import numpy as np

_CNV = np.full(256, 30, np.uint8)
_CNV[64] = _CNV[96] = 32
for _i, _c in enumerate("ACGTNRYSWKMBDHVU"):
    _CNV[ord(_c)] = _CNV[ord(_c) + 32] = _i
_CNV[128:] = _CNV[:128]


def sample_name(path):
    """file stem minus the repeated .fna/.gz/.fa/.fasta suffixes (main.cpp:108-110, application.cpp:604-630)"""
    import os
    s = os.path.basename(path)
    if "." in s:
        s = s[:s.rfind(".")]          # std::filesystem::path::stem
    while True:
        for suf in (".fna", ".gz", ".fa", ".fasta"):
            if len(s) > len(suf) and s.endswith(suf):
                s = s[:-len(suf)]
                break
        else:
            return s


def read_codes(path):
    """(contig names, symbol codes of all contigs back to back, offsets [n+1])"""
    raw = (gzip.open(path, "rb") if path.endswith(".gz") else open(path, "rb")).read()
    names, seqs = [], []
    for rec in raw.split(b">")[1:]:
        head, _, body = rec.partition(b"\n")
        head = head.rstrip(b"\r")
        b = np.frombuffer(body, np.uint8)
        if not head or not b.size:
            break
        names.append(head.decode())
        seqs.append(_CNV[b[b >= 64]])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([s.size for s in seqs])
    return names, (np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)), off
