"""BASELINE configs[1] at full size: 1000 genomes x 30 kb, 1 % SNP from one reference, defaults
(k 31, l 20, s 60000, b 50), separate files and -c; agc_amd vs the reference CLI: bytes and wall time."""
import os, sys, time, subprocess, hashlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from agc_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(2)
ref = synth.random_seq(rng, 30_000)
td = tempfile.mkdtemp(dir="/dev/shm")
files = []
genomes = [ref] + [synth.mutate(rng, ref, 0.01) for _ in range(n - 1)]
for i, g in enumerate(genomes):
    fn = os.path.join(td, f"g{i:04d}.fa")
    synth.to_fasta(fn, [g], [f"MN{i:06d}.1 synthetic genome {i}"])
    files.append(fn)
allfn = os.path.join(td, "all.fa")
synth.to_fasta(allfn, genomes[1:], [f"MN{i:06d}.1 synthetic genome {i}" for i in range(1, n)])
def run(binary, args, inputs, out):
    t0 = time.time()
    subprocess.run([binary, "create"] + args + ["-o", out] + inputs, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.time() - t0, hashlib.sha256(open(out, "rb").read()).hexdigest(), os.path.getsize(out)
REF, AMD = os.path.join(ROOT, "oracle/_ref/agc"), os.path.join(ROOT, "agc_amd/bin/agc_amd")
r = subprocess.run([os.path.join(ROOT, "agc_amd/bin/agc_amd"), "create", "-v", "1", "-t", "16", "-o", os.path.join(td, "v.agc")] + files, capture_output=True, text=True)
print(r.stderr[-400:])
for tag, args, inputs in (("files", ["-t", "16"], files), ("concat", ["-t", "16", "-c"], [files[0], allfn])):
    tr, hr, sr = run(REF, args, inputs, os.path.join(td, "r.agc"))
    ta, ha, sa = run(AMD, args, inputs, os.path.join(td, "a.agc"))
    print(f"{tag}: reference {tr:.2f} s ({n*0.03/tr:.1f} Mbp/s)  agc_amd {ta:.2f} s ({n*0.03/ta:.1f} Mbp/s)  identical={hr==ha} size={sr}", flush=True)
