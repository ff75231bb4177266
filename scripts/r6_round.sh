#!/bin/bash
# scripts/r6_round.sh PART -- the GPU runs of round 6, written under gpurun_out/r6/ on the GPU box
PART=${1:-a}
OUT=gpurun_out/r6
mkdir -p $OUT
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[1], "no line:", e); sys.exit(0)
c = d.get("config", {})
k = (d.get("roofline") or {}).get("kernels") or {}
z = c.get("zstd") or {}
print(sys.argv[1].split("/")[-1], "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "steps_only", c.get("steps_only_ms"), "close", c.get("close_ms"),
      "| zstd dev/host/wait", z.get("device_call_s"), z.get("host_pool_s"), z.get("caller_waited_s"),
      "| kernels ms:", {n: (v.get("ms_per_step") or v.get("ms_per_run")) for n, v in k.items() if isinstance(v, dict) and ("ms_per_step" in v or "ms_per_run" in v)},
      "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("archives_identical"), "fixed", c.get("fixed_cost_s"))
PY
}
bench() { # bench NAME [env...] -- the driver's command without the CPU baseline
  local name=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err; show $OUT/$name.json
}
if [ "$PART" = a ]; then
  # baseline of the round on this box: the driver's command, laps, kernel timeline
  bench a_bench_base
  AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/a_bench_laps.json 2> $OUT/a_bench_laps.txt; show $OUT/a_bench_laps.json
  ROOT=$(pwd)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/a_ktrace -o kt -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/$OUT/a_ktrace.log 2>&1)
  python scripts/step_timeline.py $(find $OUT/a_ktrace -name "*kernel_trace.csv" | head -1) > $OUT/a_step_timeline.txt 2>&1; cat $OUT/a_step_timeline.txt
  (cd /tmp && timeout 400 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $ROOT/$OUT/a_hiptrace -o ht -- python $ROOT/bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > $ROOT/$OUT/a_hiptrace.log 2>&1)
  ls -la $OUT/a_hiptrace/* | head
fi
if [ "$PART" = b ]; then
  # the pinned delta buffers with headroom; hardware queues: ROCm maps the HIP streams of a process onto GPU_MAX_HW_QUEUES (default 4) queues
  bench b_bench_fix
  bench b_bench_hwq8 GPU_MAX_HW_QUEUES=8
  bench b_bench_hwq12 GPU_MAX_HW_QUEUES=12
  bench b_bench_fix2
  bench b_bench_hwq8_2 GPU_MAX_HW_QUEUES=8
  timeout 900 python -m pytest tests/test_gpu_archive.py -m gpu -x -q -k "full_size_equals" > $OUT/b_full_identity.log 2>&1; tail -5 $OUT/b_full_identity.log
  python scripts/c3_full_identity.py 3.0 1 46b81e041ac68005a743d229bca09bf1d809d161fc3a145f6f97cba7858e6d9f > $OUT/b_full_identity_script.log 2>&1; tail -8 $OUT/b_full_identity_script.log
fi
if [ "$PART" = c ]; then
  # early collection of the whole-sample encode + the conversion queued by the compressor (agc_cmp_set_next_fasta_dev)
  timeout 600 python -m pytest tests/test_gpu_archive.py -m gpu -x -q > $OUT/c_archive_tests.log 2>&1; tail -3 $OUT/c_archive_tests.log
  bench c_bench_1
  bench c_bench_early_off AGC_AMD_EARLY_COLLECT=0
  bench c_bench_2
  AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/c_bench_laps.json 2> $OUT/c_bench_laps.txt; show $OUT/c_bench_laps.json
  python - <<'PY'
import json
for n in ("c_bench_1", "c_bench_early_off", "c_bench_2"):
    try:
        d = json.loads(open(f"gpurun_out/r6/{n}.json").read().strip().splitlines()[-1])
        print(n, d["config"]["step_ms_each_rank0"])
    except Exception as e:
        print(n, e)
PY
fi
