#!/bin/bash
OUT=gpurun_out/r3p13
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py 43000 real > $OUT/probe_base.log 2>&1
echo "base: $(grep 'run 1' $OUT/probe_base.log) $(grep -c identical $OUT/probe_base.log)"
for V in gw16 gw10 ss3 ss2; do
  AGC_HIP_LIB=$ROOT/scripts/variants/libagc_hip_$V.so AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py 43000 real > $OUT/probe_$V.log 2>&1
  echo "$V: $(grep 'run 1' $OUT/probe_$V.log) $(grep -c identical $OUT/probe_$V.log)"
done
timeout 300 python bench.py --config c1 > $OUT/bench_config_c1.json 2> $OUT/bench_config_c1.err; tail -c 900 $OUT/bench_config_c1.json
