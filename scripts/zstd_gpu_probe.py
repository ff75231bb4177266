"""Timing of agc_hip_zstd17_batch at the shape Close() produces: N packs of ~size bytes (default 50000 x ~14 KB),
against libzstd on the host threads of this box.   python scripts/zstd_gpu_probe.py [n_packs] [samples_per_pack]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agc_amd import capi
from oracle import agc_oracle as O
from tests import zstd_cases as ZC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rng = np.random.default_rng(1)
base = [ZC.delta_pack(O, rng, ns, 60000, 1e-3) for _ in range(64)]
inputs = [base[i % 64] for i in range(n)]
tot = sum(len(x) for x in inputs)
print(f"{n} packs, {tot/1e6:.1f} MB, mean {tot/n:.0f} B", flush=True)
ctx = capi.Context(0)
ctx.timing(True)
for rep in range(2):
    t = time.time()
    got = ctx.zstd17_batch(inputs)
    dt = time.time() - t
    tm = ctx.timing_get()["zstd"]
    print(f"run {rep}: wall {dt:.3f} s -> {tot/dt/1e6:.1f} MB/s; kernel {tm[0]:.1f} ms over {tm[1]} launches; out {sum(len(x) for x in got)/1e6:.1f} MB", flush=True)
    ctx.timing(True)
t = time.time()
want = [ZC.ref_frame(p) for p in base]
dt = time.time() - t
print(f"libzstd one thread: {sum(len(x) for x in base)/dt/1e6:.2f} MB/s")
assert all(got[i] == want[i % 64] for i in range(n)), "frame mismatch"
print("all frames identical to libzstd")
