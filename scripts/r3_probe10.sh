#!/bin/bash
OUT=gpurun_out/r3p10
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python scripts/r3_tiny.py 3 2 0 > $OUT/tiny.log 2>&1 || { echo "TINY FAILED"; tail -5 $OUT/tiny.log; exit 1; }
grep group $OUT/tiny.log
timeout 400 python -m pytest tests/test_gpu_zstd.py -x -q > $OUT/test_gpu_zstd.log 2>&1
tail -2 $OUT/test_gpu_zstd.log
# full packs of a collection (100 deltas, ~44 KB): group kernel (two-word records) against the one-lane kernel
for G in 3 0; do
AGC_HIP_ZSTD_GROUP=$G timeout 200 python - > $OUT/big_g$G.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from agc_amd import capi
from tests import zstd_cases as ZC
rng = np.random.default_rng(1)
data = open("scripts/data/packs_0.bin", "rb").read()
off = np.fromfile("scripts/data/packs_0.off", np.uint64)
packs = [data[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]
packs = [p for p in packs if len(p) > 2000]
inputs = []
for i in range(12000):
    s = b"".join(packs[int(x)] for x in rng.integers(0, len(packs), 11))
    a = int(rng.integers(0, 40))
    inputs.append(s[a:a + 44000])
tot = sum(len(x) for x in inputs)
ctx = capi.Context(0)
ctx.timing(True)
for rep in range(2):
    t = time.time(); got = ctx.zstd17_batch(inputs); dt = time.time() - t
    tm = ctx.timing_get()["zstd"]
    print(f"run {rep}: {len(inputs)} packs {tot/1e6:.0f} MB wall {dt:.3f} s -> {tot/dt/1e6:.1f} MB/s; kernel {tm[0]:.1f} ms", flush=True)
    ctx.timing(True)
idx = rng.integers(0, len(inputs), 60)
assert all(got[int(i)] == ZC.ref_frame(inputs[int(i)]) for i in idx)
print("sampled frames identical to libzstd")
PY
echo "44 KB packs, group=$G: $(grep 'run 1' $OUT/big_g$G.log) $(grep -c identical $OUT/big_g$G.log)"
done
