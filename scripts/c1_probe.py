"""configs[1] (1000 x 30 kb genomes) through the CLI with the start laps on stderr and a few switches flipped: where a run's wall time goes
    python scripts/c1_probe.py [n_genomes=1000]"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agc_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
rng = np.random.default_rng(2)
ref = synth.random_seq(rng, 30_000)
files = []
for i in range(n):
    g = ref if i == 0 else synth.mutate(rng, ref, 0.01)
    fn = os.path.join(td, f"g{i:04d}.fa")
    synth.to_fasta(fn, [g], [f"MN{i:06d}.1 synthetic genome {i}"])
    files.append(fn)
cli = os.path.join(ROOT, "agc_amd", "bin", "agc_amd")
def run(tag, env):
    walls = []
    for rep in range(3):
        t0 = time.perf_counter()
        r = subprocess.run([cli, "create", "-t", "16", "-v", "1", "-o", os.path.join(td, "o.agc")] + files, capture_output=True, text=True,
                           env=dict(os.environ, AGC_AMD_START_LAPS="1", **env))
        walls.append(time.perf_counter() - t0)
    print(f"== {tag}: walls {[round(w, 3) for w in walls]}")
    print("\n".join(l for l in r.stderr.splitlines() if l.startswith("start lap") or l.startswith("seconds") or l.startswith("entropy-seconds")))
run("default", {})
run("GPU_MAX_HW_QUEUES=4", {"GPU_MAX_HW_QUEUES": "4"})
run("ref store waited for", {"AGC_AMD_REF_STORE_ASYNC": "0"})
run("no early collect", {"AGC_AMD_EARLY_COLLECT": "0"})
