#!/bin/bash
# a 24-genome slice of configs[4] (5 Mbp genomes, 5 % pairwise, plasmids, -a): the archive of two ranks sharing one GPU (gloo) must be
# the single-GPU CLI's (and hence the reference's: bench.py --config c5slice compares that one) -- the device path in the N-rank
# prepare, the chunked parse and adaptive mode together, on real kernels at full contig size
OUT=gpurun_out/r5; mkdir -p $OUT
python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from agc_amd import synth
td = "/dev/shm/c5two"; os.makedirs(td, exist_ok=True)
rng = np.random.default_rng(5)
anc = synth.random_seq(rng, 5_000_000)
plasmids = [synth.random_seq(rng, int(rng.integers(20_000, 90_000))) for _ in range(12)]
with open(td + "/files.txt", "w") as fl:
    for i in range(24):
        ctg, nm = [synth.mutate(rng, anc, 0.025)], [f"chr{i}"]
        if i:
            for pi in rng.permutation(12)[: int(rng.integers(0, 3))]:
                ctg.append(synth.mutate(rng, plasmids[int(pi)], 0.025)); nm.append(f"p{i}_{int(pi)}")
        fn = f"{td}/g{i:03d}.fa"; synth.to_fasta(fn, ctg, nm); fl.write(fn + "\n")
PY
F=$(cat /dev/shm/c5two/files.txt | tr '\n' ' ')
( time agc_amd/bin/agc_amd create -a -t 16 -o /dev/shm/c5two/one.agc $F ) 2>&1 | grep real
( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 -m agc_amd.dist_create --backend gloo -a -t 16 -o /dev/shm/c5two/two.agc $F ) 2>&1 | grep -v "^$" | tail -4
sha256sum /dev/shm/c5two/one.agc /dev/shm/c5two/two.agc | tee $OUT/c5_two_ranks_check.txt
cmp /dev/shm/c5two/one.agc /dev/shm/c5two/two.agc && echo "IDENTICAL" | tee -a $OUT/c5_two_ranks_check.txt
