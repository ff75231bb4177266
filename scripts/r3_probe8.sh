#!/bin/bash
OUT=gpurun_out/r3p8
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --verify-entropy > $OUT/bench_verify_entropy.json 2> $OUT/bench_verify_entropy.err
grep "verify" $OUT/bench_verify_entropy.err | tail -3
timeout 1100 python scripts/c3_full_identity.py 3.0 5 > $OUT/c3_full_size_identity_5_samples.log 2>&1
tail -6 $OUT/c3_full_size_identity_5_samples.log
