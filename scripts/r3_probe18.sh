#!/bin/bash
OUT=gpurun_out/r3p18
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_dist_single_archive.py tests/test_gpu_archive.py -m gpu -x -q > $OUT/gpu_dist_tests.log 2>&1; tail -3 $OUT/gpu_dist_tests.log
AGC_BENCH_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/two_ranks.json 2> $OUT/two_ranks.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3p18/two_ranks.json').read().strip().splitlines()[-1]); c=d['config']
print('2 ranks on one GPU:', d['value'], c['steps_only_ms'], c['close_ms'], c['single_archive_ms_per_sample_rank0'])
PY
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3p18/bench.json').read().strip().splitlines()[-1]); c=d['config']
print('1 GPU:', d['value'], c['steps_only_ms'], c['close_ms'])
PY
