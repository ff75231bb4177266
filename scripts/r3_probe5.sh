#!/bin/bash
OUT=gpurun_out/r3p5
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
for N in 2000 9000 18000; do
AGC_HIP_LIB=$ROOT/scripts/variants/libagc_hip_prof.so AGC_HIP_ZSTD_GROUP=3 timeout 150 python scripts/zstd_gpu_probe.py $N real > $OUT/prof_$N.log 2>&1
echo "N=$N $(grep 'run 1' $OUT/prof_$N.log)"
grep zsprof $OUT/prof_$N.log | tail -2
done
