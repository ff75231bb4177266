#!/bin/bash
# bookkeeping thread on/off and frames beyond one resident round: the driver's bench shape, no CPU baseline
OUT=gpurun_out/r3p14
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python scripts/r3_tiny.py > $OUT/tiny.log 2>&1 || { echo "tiny parity check failed"; tail -5 $OUT/tiny.log; exit 1; }
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; z=c['zstd']
    print(sys.argv[2], 'value',d['value'],'steps_only_ms',c['steps_only_ms'],'close_ms',c['close_ms'],'dev_s',z['device_call_s'],'host_s',z['host_pool_s'],'dev_in',z['device_in_bytes'])
except Exception as e:
    print(sys.argv[2],'failed',e)
PY
}
run sync_book AGC_AMD_ASYNC_BOOK=0
run async_book AGC_AMD_ASYNC_BOOK=1
run extra2500 AGC_AMD_ZSTD_EXTRA_FRAMES=2500 AGC_AMD_GPU_ZSTD_SHARE=0.999
run extra5000 AGC_AMD_ZSTD_EXTRA_FRAMES=5000 AGC_AMD_GPU_ZSTD_SHARE=0.999
run extra_all AGC_AMD_ZSTD_EXTRA_FRAMES=100000 AGC_AMD_GPU_ZSTD_SHARE=0.999
run async_book2 AGC_AMD_ASYNC_BOOK=1
