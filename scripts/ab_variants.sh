#!/bin/bash
# scripts/ab_variants.sh TAG [variant.so ...] -- the step's kernel times of bench.py with alternative builds of libagc_hip.so
# (LD_PRELOAD for the host library, AGC_HIP_LIB for the ctypes binding); first entry "base" = the in-tree library
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for v in base "$@"; do
  if [ "$v" = base ]; then unset LD_PRELOAD AGC_HIP_LIB; else export LD_PRELOAD=$PWD/$v AGC_HIP_LIB=$PWD/$v; fi
  n=$(basename $v .so)
  timeout 300 python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/ab_$n.json 2> gpurun_out/$TAG/ab_$n.err
  unset LD_PRELOAD AGC_HIP_LIB
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$TAG/ab_$n.json").read().strip().splitlines()[-1])
    print("$n", "value", d["value"], "steps_only_ms", d["config"]["steps_only_ms"], "close_ms", d["config"]["close_ms"], json.dumps(d["roofline"]["kernel_ms_per_step_rank0"]))
except Exception as e:
    print("$n", "failed", e)
PY
done
