#!/bin/bash
OUT=gpurun_out/r3final3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
AGC_BENCH_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks.json 2> $OUT/bench_one_gpu_2_ranks.err; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3final3/bench_one_gpu_2_ranks.json').read().strip().splitlines()[-1]); print('2 ranks one GPU: value',d['value'],d['config']['parallelism'][:260])
except Exception as e: print('2-rank bench failed',e)
PY
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd_steps20_warmup5.json 2> $OUT/bench_driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3final3/bench_driver_cmd_steps20_warmup5.json').read().strip().splitlines()[-1])
c=d['config']; print('DRIVER CMD value',d['value'],'ms_per_step',d['ms_per_step'],'steps_only',c['steps_only_ms'],'close',c['close_ms'])
print('cpu_baseline', d.get('cpu_baseline'))
PY
