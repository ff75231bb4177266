import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agc_amd import capi, synth
from oracle import agc_oracle as O
case = sys.argv[1]
ctx = capi.Context(0)
rng = np.random.default_rng(1)
ref = synth.random_seq(rng, 5000)
ctx.ref_register(1, ref, 20)
if case == "A": text = ref.copy()
elif case == "B": text = ref[:10].copy()
elif case == "C": text = ref[:1000].copy()
elif case == "D":
    text = ref[:1000].copy(); text[500] ^= 1
elif case == "E": text = synth.mutate(rng, ref, 0.01)
elif case == "F": text = synth.random_seq(rng, 300)
mode = sys.argv[2] if len(sys.argv) > 2 else "enc"
if mode == "enc":
    enc, eoff = ctx.lz_encode_batch(text, [1], [0], [text.size])
    print(case, mode, "ok", enc.size, np.array_equal(enc, O.LZ(ref, 20).encode(text)), flush=True)
elif mode == "est":
    cost, peak = ctx.lz_estimate_batch(text, [1], [0], [text.size])
    print(case, mode, "ok", cost, peak, O.LZ(ref, 20).estimate(text, want_peak=True), flush=True)
