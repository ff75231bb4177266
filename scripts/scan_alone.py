"""agc_hip_scan_packed_dev alone on an idle GPU: a 3 Gbp sample (the bench's reference mutated at 0.1 %), the bench's splitters,
a few repetitions; prints the scan kernel's time and the number of hits (which every build must agree on).
    [AGC_HIP_LIB=agc_amd/variants/libagc_hip_X.so] python scripts/scan_alone.py [gbp=3.0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agc_amd import capi, synth_dev
gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
K, SEG = 31, 60000
dev = torch.device("cuda:0")
ref, off = synth_dev.make_reference(int(gbp * 1e9), 12345, dev)
tot = int(off[-1])
spl = synth_dev.positional_splitters(ref, off, K, SEG)
ctx = capi.Context(0)
ctx.splitters_set(spl)
codes = synth_dev.make_sample(ref, tot, 0.001, 777, dev)
pk, keep = ctx.pack_dev(codes, tot)
del codes
torch.cuda.synchronize()
ctx.timing(True)
last = 0.0
for it in range(5):
    hits = ctx.scan_packed_dev(pk, off, K, cap=1 << 20)
    ms = ctx.timing_get()["scan"][0]
    n = len(hits[0]) if isinstance(hits, tuple) else len(hits)
    print(f"scan {it}: kernel {ms - last:.3f} ms, {tot / 1e9:.2f} Gbp, {len(spl)} splitters, {n} hits, library {os.environ.get('AGC_HIP_LIB', 'in-tree')}", flush=True)
    last = ms
ctx.close()
