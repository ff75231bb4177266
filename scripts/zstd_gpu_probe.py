"""Timing of agc_hip_zstd17_batch (S3 on the GPU) against libzstd on this box's host threads.

    python scripts/zstd_gpu_probe.py [n_packs] [mode]

mode "real" (default): delta packs dumped from the pipeline (AGC_AMD_DUMP_PACKS, scripts/data/packs_0.*: the c3 twin),
  multiplied into n_packs DIFFERENT inputs (each copy starts / ends a few bytes elsewhere, several packs glued for size), so that
  the lanes of a wave parse different data as they do in Close();
mode "uniform": 64 synthetic packs repeated -- neighbouring lanes run the same control flow (the divergence-free bound)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agc_amd import capi
from tests import zstd_cases as ZC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
mode = sys.argv[2] if len(sys.argv) > 2 else "real"
rng = np.random.default_rng(1)
here = os.path.dirname(os.path.abspath(__file__))
if mode == "real" and os.path.exists(os.path.join(here, "data", "packs_0.bin")):
    data = open(os.path.join(here, "data", "packs_0.bin"), "rb").read()
    off = np.fromfile(os.path.join(here, "data", "packs_0.off"), np.uint64)
    packs = [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
    packs = [p for p in packs if len(p) > 2000]
    inputs = []
    for i in range(n):  # three packs of different groups glued (~13 KB, the size of a 20-sample pack), cut at varying places
        a, b, c = (packs[int(x)] for x in rng.integers(0, len(packs), 3))
        s = a[int(rng.integers(0, 40)):] + b + c[:len(c) - int(rng.integers(0, 40))]
        inputs.append(s[:16000])
else:
    from oracle import agc_oracle as O
    base = [ZC.delta_pack(O, rng, 32, 60000, 1e-3) for _ in range(64)]
    inputs = [base[i % 64] for i in range(n)]
tot = sum(len(x) for x in inputs)
print(f"{mode}: {n} packs, {tot/1e6:.1f} MB, mean {tot/n:.0f} B", flush=True)
ctx = capi.Context(0)
ctx.timing(True)
for rep in range(2):
    t = time.time()
    got = ctx.zstd17_batch(inputs)
    dt = time.time() - t
    tm = ctx.timing_get()["zstd"]
    print(f"run {rep}: wall {dt:.3f} s -> {tot/dt/1e6:.1f} MB/s; kernel {tm[0]:.1f} ms over {tm[1]} launches; out {sum(len(x) for x in got)/1e6:.1f} MB", flush=True)
    ctx.timing(True)
idx = rng.integers(0, n, 300)
t = time.time()
want = [ZC.ref_frame(inputs[int(i)]) for i in idx]
dt = time.time() - t
print(f"libzstd one thread: {sum(len(inputs[int(i)]) for i in idx)/dt/1e6:.2f} MB/s")
assert all(got[int(i)] == w for i, w in zip(idx, want)), "frame mismatch"
print("sampled frames identical to libzstd")
