#!/bin/bash
OUT=gpurun_out/r3p12
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python scripts/r3_tiny.py 3 2 0 > $OUT/tiny.log 2>&1 || { echo "TINY FAILED"; tail -5 $OUT/tiny.log; exit 1; }
grep group $OUT/tiny.log
for N in 36000 43000; do
  AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py $N real > $OUT/probe_g3_$N.log 2>&1
  echo "G=3 $N frames: $(grep 'run 1' $OUT/probe_g3_$N.log) $(grep -c identical $OUT/probe_g3_$N.log)"
done
timeout 400 python -m pytest tests/test_gpu_zstd.py -x -q > $OUT/test_gpu_zstd.log 2>&1
tail -2 $OUT/test_gpu_zstd.log
timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'steps_only',c['steps_only_ms'],'close',c['close_ms'],'zstd',c['zstd']['device_call_s'],c['zstd']['host_pool_s'])
PY
