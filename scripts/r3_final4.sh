#!/bin/bash
# end-of-round evidence: GPU tests, smoke, the driver's bench command, kernel trace of the same
OUT=gpurun_out/r3final4
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd_steps20_warmup5.json 2> $OUT/bench_driver_cmd.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3final4/bench_driver_cmd_steps20_warmup5.json').read().strip().splitlines()[-1])
c=d['config']; print('DRIVER CMD value',d['value'],'ms_per_step',d['ms_per_step'],'steps_only',c['steps_only_ms'],'close',c['close_ms'])
print('roofline', {k:v for k,v in d['roofline'].items() if k not in ('kernels','layout','kernel_ms_per_step_rank0')})
print('cpu_baseline', d.get('cpu_baseline'))
PY
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/ktrace -o kt -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/$OUT/ktrace.log 2>&1)
grep -E "zstd_frames_grp|lz_parse_kernel<0>|scan_packed" $OUT/ktrace/kt_kernel_stats.csv | cut -c1-160
find $OUT -name '*kernel_trace.csv' -size +8M -delete
