"""agc_hip_pack_fasta_* alone on an idle GPU: 3 Gbp of FASTA bytes in HBM -> 2-bit words, a few
repetitions, with and without other large allocations resident (the bench keeps 25 samples = ~95 GB beside it).
    python scripts/pack_alone.py [gbp=3.0] [extra_gb=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agc_amd import capi, synth_dev
gbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
extra_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda:0")
ref, off = synth_dev.make_reference(int(gbp * 1e9), 12345, dev)
tot = int(off[-1])
names = [f"chr{i + 1}" for i in range(len(off) - 1)]
ctx = capi.Context(0)
hold = [torch.empty(int(1e9), dtype=torch.uint8, device=dev) for _ in range(int(extra_gb))]
raws = []
for r in range(3):
    raw, n_raw, rb, re_ = synth_dev.make_fasta(ref, off, names, 60)
    raws.append((raw, n_raw, rb, re_))
torch.cuda.synchronize()
ctx.timing(True)
last = 0.0
for it in range(6):
    raw, n_raw, rb, re_ = raws[it % 3]
    t0 = time.perf_counter()
    pk, keep, o_ = ctx.pack_fasta_dev(raw, n_raw, rb, re_)
    wall = (time.perf_counter() - t0) * 1e3
    ms = ctx.timing_get()["pack"][0]
    print(f"pack {it}: kernel {ms - last:.3f} ms, call {wall:.2f} ms, {n_raw / 1e9:.2f} GB of FASTA, {tot / 1e9:.2f} Gbp, extra {extra_gb:.0f} GB resident", flush=True)
    last = ms
    assert np.array_equal(o_, np.asarray(off, np.uint64))
ctx.close()
