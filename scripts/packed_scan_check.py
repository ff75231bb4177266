"""3 Gbp check: scan_packed == byte scan on a bench-size sample (size-independent property of the packed path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agc_amd import capi, synth_dev
total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
dev = torch.device("cuda:0")
ref, off = synth_dev.make_reference(total, 12345, dev)
tot = int(off[-1])
spl = synth_dev.positional_splitters(ref, off, 31, 60000)
ctx = capi.Context(0)
ctx.splitters_set(spl)
smp = synth_dev.make_sample(ref, tot, 1e-3, 1000, dev)
pk, keep = ctx.pack_dev(smp, tot)
a = ctx.scan_packed_dev(pk, off, 31, cap=1 << 20)
b = ctx.scan_contigs_dev(smp.data_ptr(), off, 31, cap=1 << 20)
print("hits", a[0].size, b[0].size)
for x, y, n in zip(a, b, ("ctg", "pos", "dir", "rc")):
    if not np.array_equal(x, y):
        m = min(x.size, y.size)
        d = np.nonzero(x[:m] != y[:m])[0]
        print("MISMATCH", n, "first at", d[:5], x[d[:3]] if d.size else None, y[d[:3]] if d.size else None)
        sa, sb = set(zip(a[0].tolist(), a[1].tolist())), set(zip(b[0].tolist(), b[1].tolist()))
        print("only packed", sorted(sa - sb)[:10], "only bytes", sorted(sb - sa)[:10])
        break
else:
    print("identical")
out = torch.zeros(tot + 64, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()  # (the fill runs on torch's stream, the expansion on the library's)
ctx.expand_dev(pk, out.data_ptr())
print("expand equal:", bool(torch.equal(out[:tot], smp[:tot])))
