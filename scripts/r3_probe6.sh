#!/bin/bash
OUT=gpurun_out/r3p6
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 120 python scripts/r3_tiny.py 3 2 0 > $OUT/tiny.log 2>&1 || { echo "TINY FAILED"; tail -5 $OUT/tiny.log; exit 1; }
grep group $OUT/tiny.log
AGC_HIP_LIB=$ROOT/scripts/variants/libagc_hip_prof.so AGC_HIP_ZSTD_GROUP=3 timeout 150 python scripts/zstd_gpu_probe.py 36000 real > $OUT/prof.log 2>&1
echo "prof: $(grep 'run 1' $OUT/prof.log)"; grep zsprof $OUT/prof.log | tail -2
for N in 36000 43000; do
  AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py $N real > $OUT/probe_g3_$N.log 2>&1
  echo "G=3 $N frames: $(grep 'run 1' $OUT/probe_g3_$N.log) $(grep -c identical $OUT/probe_g3_$N.log)"
done
timeout 300 python -m pytest tests/test_gpu_zstd.py -x -q > $OUT/test_gpu_zstd.log 2>&1
tail -2 $OUT/test_gpu_zstd.log
