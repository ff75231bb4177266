#!/bin/bash
# round 3: the evidence kept under profiles/r3/ (run on the GPU box; copy gpurun_out/r3final/* into profiles/r3/)
OUT=gpurun_out/r3final
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd_steps20_warmup5.json 2> $OUT/bench_driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3final/bench_driver_cmd_steps20_warmup5.json').read().strip().splitlines()[-1])
c=d['config']; print('DRIVER CMD value',d['value'],'ms_per_step',d['ms_per_step'],'steps_only',c['steps_only_ms'],'close',c['close_ms'])
print('roofline', {k:v for k,v in d['roofline'].items() if k not in ('kernels','layout','kernel_ms_per_step_rank0')})
print('cpu_baseline', d.get('cpu_baseline'))
PY
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "import json;d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]);print('DEFAULT value',d['value'])"
timeout 900 bash scripts/profile_round.sh r3 > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log
timeout 300 python bench.py --from-fasta 2 > $OUT/bench_from_fasta.json 2> $OUT/bench_from_fasta.err; tail -c 600 $OUT/bench_from_fasta.json
AGC_BENCH_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks.json 2> $OUT/bench_one_gpu_2_ranks.err; tail -c 400 $OUT/bench_one_gpu_2_ranks.json
timeout 500 python scripts/fuzz_archives.py --from 12000 --count 60 > $OUT/fuzz_gpu_60_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_60_cases.log
