"""Kernel-level measurement at (scaled) human size: scan + encode of both-splitter segments."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agc_amd import capi, synth_dev
total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000_000
d = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
k, mml, seg = 31, 15, 60000
dev = torch.device("cuda:0")
t0 = time.time()
ref, off = synth_dev.make_reference(total, 12345, dev)
tot = int(off[-1])
spl = synth_dev.positional_splitters(ref, off, k, seg)
torch.cuda.synchronize()
print(f"gen ref {tot/1e9:.2f} Gbp, {spl.size} splitters, {time.time()-t0:.1f}s", flush=True)
ctx = capi.Context(0)
ctx.timing(True)
ctx.splitters_set(spl)
t = time.time(); ctg, pos, hd, hr = ctx.scan_contigs_dev(ref.data_ptr(), off, k, cap=1 << 20); t_scan = time.time() - t
print(f"ref scan: {pos.size} hits, wall {t_scan*1e3:.1f} ms", ctx.timing_get()["scan"], flush=True)

def segments(ctg, pos, hd, hr, off):
    """both-splitter segments only: (abs start, len, kfront, kback) canonical"""
    can = np.minimum(hd, hr)
    same = ctg[1:] == ctg[:-1]
    i = np.nonzero(same)[0]
    start = off[ctg[i]].astype(np.int64) + pos[i].astype(np.int64) + 1 - k
    ln = pos[i + 1].astype(np.int64) - pos[i].astype(np.int64) + k
    return start, ln, can[i], can[i + 1]

start, ln, kf, kb = segments(ctg, pos, hd, hr, off)
rc = (kf >= kb).astype(np.uint8)
pk = np.stack([np.minimum(kf, kb), np.maximum(kf, kb)], 1)
# unique pk -> gid
_, first_idx = np.unique(pk, axis=0, return_index=True)
first_idx.sort()
gids = 16 + np.arange(first_idx.size)
pkmap = {(int(a), int(b)): int(g) for (a, b), g in zip(pk[first_idx], gids)}
t = time.time()
ctx.ref_register_batch_dev(gids, ref.data_ptr(), start[first_idx].astype(np.uint64), ln[first_idx].astype(np.uint32), rc[first_idx], mml)
print(f"registered {gids.size} refs in {time.time()-t:.2f}s", {n: v for n, v in ctx.timing_get().items() if v[1]}, flush=True)
for s in range(2):
    t = time.time(); smp = synth_dev.make_sample(ref, tot, d, 100 + s, dev); torch.cuda.synchronize()
    print(f"gen sample {time.time()-t:.1f}s", flush=True)
    ctx.timing(True)
    t = time.time(); c2, p2, d2, r2 = ctx.scan_contigs_dev(smp.data_ptr(), off, k, cap=1 << 20); t_scan = time.time() - t
    st, l2, f2, b2 = segments(c2, p2, d2, r2, off)
    rc2 = (f2 >= b2).astype(np.uint8)
    t = time.time()
    g2 = np.array([pkmap.get((int(min(a, b)), int(max(a, b))), -1) for a, b in zip(f2, b2)], np.int64)
    t_cls = time.time() - t
    known = g2 >= 0
    t = time.time()
    enc, eoff = ctx.lz_encode_batch_dev(smp.data_ptr(), g2[known].astype(np.uint32), st[known].astype(np.uint64), l2[known].astype(np.uint32), rc2[known])
    t_enc = time.time() - t
    tm = ctx.timing_get()
    nb = int(l2[known].sum())
    print(f"sample {s}: hits {p2.size} segs {st.size} known {int(known.sum())} ({nb/1e9:.3f} Gbp) delta {enc.size/1e6:.2f} MB | "
          f"scan wall {t_scan*1e3:.1f} ms kern {tm['scan'][0]:.2f} ms -> {tot/tm['scan'][0]/1e6:.1f} Gbp/s | classify(py) {t_cls*1e3:.0f} ms | "
          f"encode wall {t_enc*1e3:.1f} ms kern {tm['encode'][0]:.2f} ms revcomp {tm['revcomp'][0]:.2f} ms -> {nb/tm['encode'][0]/1e6:.1f} Gbp/s", flush=True)
