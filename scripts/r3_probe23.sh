#!/bin/bash
OUT=gpurun_out/r3p23
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_archive.py -m gpu -x -q > $OUT/gpu_archive_tests.log 2>&1; tail -2 $OUT/gpu_archive_tests.log
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; z=c['zstd']
    print(sys.argv[2], 'value',d['value'],'steps_only_ms',c['steps_only_ms'],'close_ms',c['close_ms'],'dev_s',z['device_call_s'],'host_s',z['host_pool_s'], 'delta_bytes', c['delta_bytes_per_step'])
except Exception as e:
    print(sys.argv[2],'failed',e)
PY
}
for i in 1 2 3; do
  run sync_enc_$i AGC_AMD_ASYNC_ENCODE=0
  run async_enc_$i AGC_AMD_ASYNC_ENCODE=1
done
run laps_async AGC_AMD_ASYNC_ENCODE=1 AGC_AMD_LAPS=1
