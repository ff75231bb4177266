#!/bin/bash
OUT=gpurun_out/r3p24
mkdir -p $OUT
export TMPDIR=/tmp
SHA=87fa6261593cd046a4ec7ecc8ec216686e6d82cd39b62edb2c031141e83517fa
timeout 300 python scripts/c3_full_identity.py 3.0 5 $SHA > $OUT/identity_async.log 2>&1; tail -3 $OUT/identity_async.log
AGC_IDENTITY_ANNOUNCE=1 timeout 300 python scripts/c3_full_identity.py 3.0 5 $SHA > $OUT/identity_async_announce.log 2>&1; tail -3 $OUT/identity_async_announce.log
AGC_AMD_ASYNC_ENCODE=0 AGC_AMD_ASYNC_BOOK=0 timeout 300 python scripts/c3_full_identity.py 3.0 5 $SHA > $OUT/identity_sync.log 2>&1; tail -1 $OUT/identity_sync.log
for m in 1 0; do
  AGC_BENCH_ARCHIVE=/tmp/bench_$m.agc AGC_AMD_ASYNC_ENCODE=$m AGC_AMD_ASYNC_BOOK=$m timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_archive_$m.json 2> $OUT/bench_archive_$m.err
  python -c "
import json
d=json.loads(open('$OUT/bench_archive_$m.json').read().strip().splitlines()[-1]); c=d['config']
print('async=$m', d['value'], c['steps_only_ms'], c['close_ms'], c.get('archive_bytes'), c.get('archive_sha256'))"
done
