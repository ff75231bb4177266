"""Step-by-step GPU probe with flushed progress lines (debug aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t0 = time.time()
def say(*a):
    print(f"[{time.time()-t0:7.2f}]", *a, flush=True)
from agc_amd import capi, synth
from oracle import agc_oracle as O
say("imports done")
ctx = capi.Context(0)
say("context created")
rng = np.random.default_rng(1)
ref = synth.random_seq(rng, 5000)
text = synth.mutate(rng, ref, 0.01)
ctx.ref_register(1, ref, 20)
say("ref registered")
tab, is16 = ctx.ref_index_get(1)
say("index fetched", tab.size, is16, "match:", np.array_equal(tab, O.LZ(ref, 20).index().astype(np.uint32)))
enc, eoff = ctx.lz_encode_batch(text, [1], [0], [text.size])
say("encoded", enc.size, "match:", np.array_equal(enc, O.LZ(ref, 20).encode(text)))
cost, peak = ctx.lz_estimate_batch(text, [1], [0], [text.size])
say("estimate", cost, peak, O.LZ(ref, 20).estimate(text, want_peak=True))
cv = ctx.lz_cost_vector_batch(text, [1], [0], [text.size], None, [1])
say("costvec match:", np.array_equal(cv, O.LZ(ref, 20).cost_vector(text, 1)))
k = 21
spl = O.determine_splitters([ref], k, 500)
ctx.splitters_set(spl)
say("splitters set", spl.size)
got = ctx.scan_contigs(text, [0, text.size], k)
s = O.scan_contig(text, k, spl)
m = s["back_full"] == 1
say("scan", got[1][:5], (s["start"][m] + s["len"][m] - 1)[:5], "match:", np.array_equal(got[1], (s["start"][m] + s["len"][m] - 1)))
say("done")
