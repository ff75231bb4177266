#!/bin/bash
OUT=gpurun_out/r3p11
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_lz.py -x -q > $OUT/test_gpu_lz.log 2>&1
tail -2 $OUT/test_gpu_lz.log
AGC_AMD_LAPS=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']; print('value',d['value'],'steps_only',c['steps_only_ms'],'close',c['close_ms'],'kernels',d['roofline']['kernel_ms_per_step_rank0'])
PY
grep "lap splitpoints" $OUT/bench.err | awk '{print $3}' | tr '\n' ' '
