"""Full-pipeline probe at scaled human size: stage times of the host compressor."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from agc_amd import host, synth_dev
total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000_000
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 3
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 64
d = 1e-3
k, mml, seg, pack = 31, 15, 60000, 100
dev = torch.device("cuda:0")
ref, off = synth_dev.make_reference(total, 12345, dev)
tot = int(off[-1])
spl = synth_dev.positional_splitters(ref, off, k, seg)
names = [f"chr{i+1}" for i in range(len(off) - 1)]
c = host.Compressor(0)
c.create("", pack, k, None, seg, mml, n_threads=threads)
c.set_splitters(spl)
print("zstd", c.zstd_version(), flush=True)
def show(tag, t, prev):
    s = c.stats()
    d_ = {k_: (s[k_] - prev.get(k_, 0)) for k_ in s}
    print(f"{tag}: wall {t*1e3:.1f} ms | " + " ".join(f"{k_}={d_[k_]:.3f}" if k_.startswith("t_") else f"{k_}={int(d_[k_])}" for k_ in s if d_[k_]), flush=True)
    return s
prev = {}
t = time.time(); c.add_sample_dev("ref", names, ref.data_ptr(), off); prev = show("reference sample", time.time() - t, prev)
for s in range(n_samples):
    smp = synth_dev.make_sample(ref, tot, d, 100 + s, dev); torch.cuda.synchronize()
    t = time.time(); c.add_sample_dev(f"s{s}", names, smp.data_ptr(), off); dt = time.time() - t
    prev = show(f"sample {s} ({tot/dt/1e9:.1f} Gbp/s)", dt, prev)
t = time.time(); c.close(threads); prev = show("close", time.time() - t, prev)
