#!/bin/bash
# every LZ launch of a small slice of configs[4] on stderr (AGC_HIP_CHUNK_LOG): texts, sizes, chunked or not, kernel times
OUT=gpurun_out/r5; mkdir -p $OUT
python - <<'PY'
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
from agc_amd import synth
td = "/dev/shm/c5log"; os.makedirs(td, exist_ok=True)
rng = np.random.default_rng(5)
anc = synth.random_seq(rng, 5_000_000)
files = []
plasmids = [synth.random_seq(rng, int(rng.integers(20_000, 90_000))) for _ in range(12)]
for i in range(int(os.environ.get("C5N", "32"))):
    ctg, nm = [synth.mutate(rng, anc, 0.025)], [f"chr{i}"]
    if i:
        for pi in rng.permutation(12)[: int(rng.integers(0, 3))]:
            ctg.append(synth.mutate(rng, plasmids[int(pi)], 0.025)); nm.append(f"p{i}_{int(pi)}")
    fn = os.path.join(td, f"g{i}.fa"); synth.to_fasta(fn, ctg, nm); files.append(fn)
env = dict(os.environ, AGC_HIP_CHUNK_LOG="1")
r = subprocess.run(["agc_amd/bin/agc_amd", "create", "-a", "-v", "1", "-t", "16", "-o", td + "/o.agc"] + files, capture_output=True, text=True, env=env)
open("gpurun_out/r5/e_c5_chunk_log.txt", "w").write(r.stderr)
PY
grep -c "launch mode" $OUT/e_c5_chunk_log.txt; grep -A20 "launch mode 0" $OUT/e_c5_chunk_log.txt | awk '/hop kernel/{split($0,a,"hop kernel "); split(a[2],b," "); show=(b[1]+0>5)} show' | head -120; grep "^bases\|^seconds" $OUT/e_c5_chunk_log.txt
