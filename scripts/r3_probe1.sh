#!/bin/bash
# round 3: parity of the group kernels (tiny case first, short timeouts), then launch times of the zstd stage per variant
OUT=gpurun_out/r3p1
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python scripts/r3_tiny.py 3 2 0 > $OUT/tiny.log 2>&1 || { echo "TINY FAILED"; tail -5 $OUT/tiny.log; exit 1; }
cat $OUT/tiny.log | grep group
timeout 400 python -m pytest tests/test_gpu_zstd.py -x -q > $OUT/test_gpu_zstd.log 2>&1
tail -3 $OUT/test_gpu_zstd.log
for G in 3 2 0; do
  AGC_HIP_ZSTD_GROUP=$G timeout 120 python scripts/zstd_gpu_probe.py 36000 real > $OUT/probe_g$G.log 2>&1
  echo "G=$G: $(grep 'run 1' $OUT/probe_g$G.log) $(grep -c identical $OUT/probe_g$G.log)"
done
for W in 8 10; do
  AGC_HIP_LIB=$(pwd)/scripts/variants/libagc_hip_w$W.so AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py 36000 real > $OUT/probe_g3_w$W.log 2>&1
  echo "G=3 walk=$W: $(grep 'run 1' $OUT/probe_g3_w$W.log) $(grep -c identical $OUT/probe_g3_w$W.log)"
done
AGC_HIP_ZSTD_GROUP=3 AGC_HIP_ZSTD_DEBUG=1 timeout 120 python scripts/zstd_gpu_probe.py 36000 real > $OUT/probe_g3_parseonly.log 2>&1
echo "G=3 parse only: $(grep 'run 1' $OUT/probe_g3_parseonly.log)"
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
tail -c 1200 $OUT/bench_steps20.json
