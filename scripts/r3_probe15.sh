#!/bin/bash
# A/B of the bookkeeping thread: alternating runs + one lap profile each
OUT=gpurun_out/r3p15
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; z=c['zstd']
    print(sys.argv[2], 'value',d['value'],'steps_only_ms',c['steps_only_ms'],'close_ms',c['close_ms'],'dev_s',z['device_call_s'],'host_s',z['host_pool_s'])
except Exception as e:
    print(sys.argv[2],'failed',e)
PY
}
for i in 1 2 3 4; do
  run sync_$i AGC_AMD_ASYNC_BOOK=0
  run async_$i AGC_AMD_ASYNC_BOOK=1
done
run laps_sync AGC_AMD_ASYNC_BOOK=0 AGC_AMD_LAPS=1
run laps_async AGC_AMD_ASYNC_BOOK=1 AGC_AMD_LAPS=1
