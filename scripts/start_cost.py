"""where the fixed cost of a one-contig `agc_amd create` goes: wall time of every call of the host API on a 2 kb contig
    python scripts/start_cost.py"""
import os, sys, time, tempfile
t00 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agc_amd import synth
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fn = os.path.join(td, "tiny.fa")
synth.to_fasta(fn, [synth.random_seq(np.random.default_rng(1), 2000)], ["tiny"])
import subprocess
for rep in range(3):
    t0 = time.perf_counter()
    subprocess.run([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "agc_amd", "bin", "agc_amd"), "create", "-o", os.path.join(td, "t.agc"), fn], check=True, capture_output=True)
    print(f"CLI wall {time.perf_counter() - t0:.3f} s")
import ctypes
t0 = time.perf_counter()
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
n = ctypes.c_int()
hip.hipGetDeviceCount(ctypes.byref(n))
print(f"dlopen libamdhip64 + hipGetDeviceCount {time.perf_counter() - t0:.3f} s")
t0 = time.perf_counter()
hip.hipFree(None)
print(f"hipFree(0): runtime / context init {time.perf_counter() - t0:.3f} s")
from agc_amd import host
t0 = time.perf_counter()
L = host.load()
print(f"load libagc_host.so + libagc_hip.so (code objects registered) {time.perf_counter() - t0:.3f} s")
for rep in range(2):
    t0 = time.perf_counter()
    c = host.Compressor(0)
    t1 = time.perf_counter()
    c.create(os.path.join(td, "x.agc"), 50, 31, fn, 60000, 20, n_threads=16)
    t2 = time.perf_counter()
    c.add_sample_files([("tiny", fn)], 16)
    t3 = time.perf_counter()
    c.close(16)
    t4 = time.perf_counter()
    c.close_handle()
    t5 = time.perf_counter()
    print(f"rep {rep}: Compressor() {t1 - t0:.3f}  create (zstd dlopen, hip ctx, pools, reference file, determine_splitters) {t2 - t1:.3f}  add_sample_files {t3 - t2:.3f}  close {t4 - t3:.3f}  destroy {t5 - t4:.3f}")
