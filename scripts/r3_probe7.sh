#!/bin/bash
OUT=gpurun_out/r3p7
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python scripts/r3_tiny.py 3 0 > $OUT/tiny.log 2>&1 || { echo "TINY FAILED"; tail -5 $OUT/tiny.log; exit 1; }
timeout 300 python -m pytest tests/test_gpu_zstd.py -x -q > $OUT/test_gpu_zstd.log 2>&1
tail -2 $OUT/test_gpu_zstd.log
AGC_AMD_LAPS=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3p7/bench_steps20.json').read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'ms_per_step',d['ms_per_step'],'steps_only',c['steps_only_ms'],'close',c['close_ms'],'zstd',c['zstd'])
PY
grep -i "entropy" $OUT/bench_steps20.err | tail -7
