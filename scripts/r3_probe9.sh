#!/bin/bash
OUT=gpurun_out/r3p9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_scan.py -x -q > $OUT/test_gpu_scan.log 2>&1
tail -3 $OUT/test_gpu_scan.log
for P in "" "--no-prefetch"; do
AGC_AMD_LAPS=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $P > $OUT/bench$P.json 2> $OUT/bench$P.err
python - "$OUT/bench$P.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c=d['config']; print(sys.argv[1],'value',d['value'],'steps_only',c['steps_only_ms'],'close',c['close_ms'],'kernels',d['roofline']['kernel_ms_per_step_rank0'])
PY
done
grep "lap scan\|lap encode \|lap splitpoints" $OUT/bench.err | tail -6
