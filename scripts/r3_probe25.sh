#!/bin/bash
OUT=gpurun_out/r3p25
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python scripts/r3_tiny.py > $OUT/tiny.log 2>&1 || { echo "tiny parity check failed"; tail -5 $OUT/tiny.log; exit 1; }
for i in 1 2 3; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b$i.json 2> $OUT/b$i.err
  python -c "
import json
d=json.loads(open('$OUT/b$i.json').read().strip().splitlines()[-1]); c=d['config']; z=c['zstd']
print('run $i', d['value'], c['steps_only_ms'], c['close_ms'], 'dev', z['device_call_s'], 'host', z['host_pool_s'])"
done
