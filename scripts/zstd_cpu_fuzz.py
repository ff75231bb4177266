"""Bulk CPU parity of the zstd level-17 encoder (agc_amd/csrc/zstd/*.h, host build of tests/zstd_host) with the image's libzstd:

    python scripts/zstd_cpu_fuzz.py [count] [seed]

Inputs: the real delta packs dumped from the pipeline (scripts/data/packs_0.*, when present) glued and cut at random places to
sizes from 1 byte to 128 KiB, plus the synthetic corpus of tests/zstd_cases.py.  Every frame of the micro-step parser (the form
the kernel runs) is compared byte for byte with ZSTD_compressCCtx(level 17)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from tests import zstd_cases as ZC
from tests.zstd_host import build as zbuild
from oracle import agc_oracle as O

count = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
H = C.CDLL(zbuild.build())
H.zs_host_compress2.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
H.zs_host_compress2.restype = C.c_uint32
here = os.path.dirname(os.path.abspath(__file__))
inputs = []
pk = os.path.join(here, "data", "packs_0.bin")
if os.path.exists(pk):
    data = open(pk, "rb").read()
    off = np.fromfile(os.path.join(here, "data", "packs_0.off"), np.uint64)
    packs = [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    for i in range(count * 3 // 4):
        k = int(rng.integers(1, 12))
        s = b"".join(packs[int(x)] for x in rng.integers(0, len(packs), k))
        a = int(rng.integers(0, 64))
        n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 16384), rng.integers(16384, 16400), rng.integers(16384, 131072)], p=[0.1, 0.6, 0.05, 0.25]))
        inputs.append(s[a:a + n])
inputs += ZC.corpus(O, seed + 77, count - len(inputs))
bad = 0
tot = 0
t0 = time.time()
for i, p in enumerate(inputs):
    n = len(p)
    if n == 0:
        continue
    cp = np.array(ZC.ref_cparams(n), np.uint32)
    out = np.zeros(n + 64, np.uint8)
    k = H.zs_host_compress2(bytes(p), n, cp.ctypes.data, out.ctypes.data, 0)
    if out[:k].tobytes() != ZC.ref_frame(p):
        bad += 1
        print("MISMATCH input", i, "size", n, flush=True)
    tot += n
print(f"{len(inputs)} inputs, {tot / 1e6:.1f} MB, libzstd {ZC.libzstd().ZSTD_versionNumber()}, mismatches: {bad}, {time.time() - t0:.0f} s")
