#!/bin/bash
OUT=gpurun_out/r3p4
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 120 python scripts/r3_tiny.py 3 2 0 > $OUT/tiny.log 2>&1 || { echo "TINY FAILED"; tail -5 $OUT/tiny.log; exit 1; }
grep group $OUT/tiny.log
timeout 300 python -m pytest tests/test_gpu_zstd.py -x -q > $OUT/test_gpu_zstd.log 2>&1
tail -2 $OUT/test_gpu_zstd.log
for N in 36000 42000; do
  AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py $N real > $OUT/probe_g3_$N.log 2>&1
  echo "G=3 $N frames: $(grep 'run 1' $OUT/probe_g3_$N.log) $(grep -c identical $OUT/probe_g3_$N.log)"
done
for W in 8 16; do
  AGC_HIP_LIB=$ROOT/scripts/variants/libagc_hip_gw$W.so AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py 36000 real > $OUT/probe_g3_w$W.log 2>&1
  echo "G=3 W=$W: $(grep 'run 1' $OUT/probe_g3_w$W.log) $(grep -c identical $OUT/probe_g3_w$W.log)"
done
AGC_AMD_LAPS=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3p4/bench_steps20.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'ms_per_step',d['ms_per_step'],'steps_only',c['steps_only_ms'],'close',c['close_ms'],'zstd',c['zstd'])
PY
grep -i "entropy" $OUT/bench_steps20.err | tail -12
