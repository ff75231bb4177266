"""a handful of frames through the group kernels, compared with libzstd (first thing to run on the GPU after a kernel change)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agc_amd import capi
from tests import zstd_cases as ZC
from oracle import agc_oracle as O
rng = np.random.default_rng(3)
inputs = [ZC.delta_pack(O, rng, n, 60000, 1e-3)[:16384] for n in (2, 5, 25, 30)] + [b"A" * 5000, b"ABC" * 2000, bytes(rng.integers(65, 69, 3000, dtype=np.uint8))]
ctx = capi.Context(0)
for g in sys.argv[1:] or ["3"]:
    os.environ["AGC_HIP_ZSTD_GROUP"] = g
    t = time.time()
    got = ctx.zstd17_batch(inputs)
    ok = all(a == ZC.ref_frame(p) for a, p in zip(got, inputs))
    print(f"group {g}: {len(inputs)} frames in {time.time() - t:.2f} s, identical to libzstd: {ok}", flush=True)
    assert ok
