#!/bin/bash
OUT=gpurun_out/r3p26
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 python scripts/r3_tiny.py > $OUT/tiny.log 2>&1 || { echo "tiny parity check failed"; tail -5 $OUT/tiny.log; exit 1; }
timeout 400 python -m pytest tests/test_gpu_zstd.py -m gpu -x -q > $OUT/gpu_zstd_tests.log 2>&1; tail -3 $OUT/gpu_zstd_tests.log
for refs in 0 512; do
  AGC_AMD_GPU_ZSTD_REFS=$refs AGC_AMD_LAPS=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_refs_$refs.json 2> $OUT/bench_refs_$refs.err
  python -c "
import json
d=json.loads(open('$OUT/bench_refs_$refs.json').read().strip().splitlines()[-1]); c=d['config']
print('refs_min=$refs', d['value'], c['steps_only_ms'], c['close_ms'], c['setup_not_timed'])"
  grep -E "entropy jobs|entropy lap (host pool|wait for the device)" $OUT/bench_refs_$refs.err | head -6
done
