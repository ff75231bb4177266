#!/bin/bash
OUT=gpurun_out/r3p22
mkdir -p $OUT
export TMPDIR=/tmp
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
run g3 AGC_HIP_ZSTD_GROUP=3
run g2_all AGC_HIP_ZSTD_GROUP=2 AGC_AMD_ZSTD_EXTRA_FRAMES=100000 AGC_AMD_GPU_ZSTD_SHARE=0.999
run g2_5000 AGC_HIP_ZSTD_GROUP=2 AGC_AMD_ZSTD_EXTRA_FRAMES=5000 AGC_AMD_GPU_ZSTD_SHARE=0.999
run g2_split AGC_HIP_ZSTD_GROUP=2
