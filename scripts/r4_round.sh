#!/bin/bash
# scripts/r4_round.sh PART -- the end-of-round evidence of round 4, written under gpurun_out/r4/ on the GPU box
#   PART=a: GPU tests, smoke, the driver's bench command (+ default, configs c1 / c4twin / c5twin, two ranks on one GPU),
#           full-size identity of BASELINE configs[2] x 5 samples against the reference CLI
#   PART=b: rocprofv3 passes (kernel trace + stats, FETCH_SIZE, WRITE_SIZE, calibration kernels, SQ counters of the LZ kernels)
PART=${1:-a}
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get("config", {})
print(sys.argv[1].split("/")[-1], "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "steps_only", c.get("steps_only_ms"), "close", c.get("close_ms"))
PY
}
if [ "$PART" = a ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd_steps20_warmup5.json 2> $OUT/bench_driver_cmd.err; show $OUT/bench_driver_cmd_steps20_warmup5.json
  timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_default_no_cpu_baseline.json 2> /dev/null; show $OUT/bench_default_no_cpu_baseline.json
  for i in 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd_steps20_warmup5_run$i.json 2>/dev/null; show $OUT/bench_driver_cmd_steps20_warmup5_run$i.json; done
  for cfg in c1 c4twin c5twin; do timeout 600 python bench.py --config $cfg > $OUT/bench_config_$cfg.json 2> $OUT/bench_config_$cfg.err; tail -c 600 $OUT/bench_config_$cfg.json; echo; done
  AGC_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks.json 2> $OUT/bench_one_gpu_2_ranks.err; show $OUT/bench_one_gpu_2_ranks.json
  AGC_BENCH_ONE_GPU=1 AGC_BENCH_SERIAL_PREPARE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks_serial_prepare.json 2> $OUT/bench_one_gpu_2_ranks_serial_prepare.err; show $OUT/bench_one_gpu_2_ranks_serial_prepare.json
  timeout 900 python scripts/c3_full_identity.py 3.0 5 87fa6261593cd046a4ec7ecc8ec216686e6d82cd39b62edb2c031141e83517fa > $OUT/c3_full_size_identity_5_samples.log 2>&1; tail -4 $OUT/c3_full_size_identity_5_samples.log
else
  bash scripts/profile_round.sh r4
  LZ_PMC_KERNELS="lz_parse_kernel|scan_packed|key_filter|split_point|idx_|known_|group_lookup" bash scripts/lz_pmc_probe.sh r4 > $OUT/lz_pmc_probe.log 2>&1; tail -5 $OUT/lz_pmc_probe.log
  AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/bench_laps.txt; grep -c lap $OUT/bench_laps.txt
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --verify-entropy > $OUT/bench_verify_entropy.json 2> $OUT/bench_verify_entropy.log; grep -h "verify" $OUT/bench_verify_entropy.log | tail -3
fi
ls $OUT | head -50
