#!/bin/bash
# scripts/r5_round.sh PART -- the GPU runs of round 5, written under gpurun_out/r5/ on the GPU box
#   PART=a: the new one-pass pack kernel's tests first (under a short timeout: a look-back that never ends must not hold the box),
#           then the GPU test suite, the driver's bench command (FASTA bytes in HBM) and its --prepacked control, configs c1 / c5twin / c5slice
#   PART=b: rocprofv3 passes (kernel trace + stats, FETCH_SIZE, WRITE_SIZE) of the bench and of the divergent configs, laps, --verify-entropy
#   PART=c: end of round: tests, smoke, bench lines, two ranks on one GPU, fuzz through the real kernels
PART=${1:-a}
OUT=gpurun_out/r5
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
print(sys.argv[1].split("/")[-1], "value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "steps_only", c.get("steps_only_ms"), "close", c.get("close_ms"),
      "| kernels ms:", {n: (v.get("ms_per_step") or v.get("ms_per_run")) for n, v in k.items() if isinstance(v, dict) and ("ms_per_step" in v or "ms_per_run" in v)},
      "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("archives_identical"), "fixed", c.get("fixed_cost_s"))
PY
}
if [ "$PART" = a ]; then
  timeout 300 python -m pytest tests/test_gpu_scan.py -m gpu -x -q -k "pack_fasta" > $OUT/pack_fasta_tests.log 2>&1; PF=$?; tail -15 $OUT/pack_fasta_tests.log
  timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_scan.py::test_pack_fasta_matches_oracle_preprocess > $OUT/gpu_tests.log 2>&1; tail -12 $OUT/gpu_tests.log
  if [ $PF = 0 ]; then
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd_steps20_warmup5_no_cpu_baseline.json 2> $OUT/bench_driver_cmd.err; show $OUT/bench_driver_cmd_steps20_warmup5_no_cpu_baseline.json; tail -3 $OUT/bench_driver_cmd.err
  fi
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prepacked > $OUT/bench_prepacked_steps20_warmup5.json 2> $OUT/bench_prepacked.err; show $OUT/bench_prepacked_steps20_warmup5.json
  for cfg in c1 c5twin; do timeout 600 python bench.py --config $cfg > $OUT/bench_config_$cfg.json 2> $OUT/bench_config_$cfg.err; show $OUT/bench_config_$cfg.json; done
  timeout 900 python bench.py --config c5slice --c5-samples ${C5N:-128} > $OUT/bench_config_c5slice.json 2> $OUT/bench_config_c5slice.err; show $OUT/bench_config_c5slice.json; tail -3 $OUT/bench_config_c5slice.err
fi
if [ "$PART" = b ]; then
  timeout 300 python -m pytest tests/test_gpu_scan.py tests/test_dist_single_archive.py -m gpu -x -q -k "pack_fasta or on_the_gpu" > $OUT/b_pack_and_dist_tests.log 2>&1; tail -4 $OUT/b_pack_and_dist_tests.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b_bench_driver_cmd_no_cpu_baseline.json 2> $OUT/b_bench_driver_cmd.err; show $OUT/b_bench_driver_cmd_no_cpu_baseline.json; tail -2 $OUT/b_bench_driver_cmd.err
  for cfg in c5twin c1; do timeout 600 python bench.py --config $cfg > $OUT/b_bench_config_$cfg.json 2> $OUT/b_bench_config_$cfg.err; show $OUT/b_bench_config_$cfg.json; grep -h "^bases" $OUT/b_bench_config_$cfg.err | tail -1; done
  timeout 900 python bench.py --config c5slice --c5-samples ${C5N:-128} > $OUT/b_bench_config_c5slice.json 2> $OUT/b_bench_config_c5slice.err; show $OUT/b_bench_config_c5slice.json; grep -h "^bases\|^seconds" $OUT/b_bench_config_c5slice.err | tail -2
  AGC_BENCH_ONE_GPU=1 AGC_BENCH_SERIAL_PREPARE=1 AGC_AMD_LAPS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/b_bench_one_gpu_2_ranks_serial_prepare.json 2> $OUT/b_bench_one_gpu_2_ranks_serial_prepare.err; show $OUT/b_bench_one_gpu_2_ranks_serial_prepare.json
  python - $OUT/b_bench_one_gpu_2_ranks_serial_prepare.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ms per sample rank0:", d["config"].get("single_archive_ms_per_sample_rank0"))
    print(d["config"].get("parallelism"))
except Exception as e:
    print("no line", e)
PY
  tail -5 $OUT/b_bench_one_gpu_2_ranks_serial_prepare.err
fi
if [ "$PART" = c ]; then
  timeout 200 python -m pytest tests/test_gpu_scan.py -m gpu -x -q -k "pack_fasta" > $OUT/c_pack_tests.log 2>&1; tail -3 $OUT/c_pack_tests.log
  timeout 300 python -m pytest tests/test_gpu_lz.py -m gpu -x -q -k "not every_launch" > $OUT/c_lz_tests.log 2>&1; tail -15 $OUT/c_lz_tests.log
  timeout 900 python -m pytest tests/test_gpu_lz.py -m gpu -x -q -k "every_launch" > $OUT/c_lz_forced_chunks.log 2>&1; tail -25 $OUT/c_lz_forced_chunks.log
  timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/c_bench_steps10.json 2> $OUT/c_bench_steps10.err; show $OUT/c_bench_steps10.json
  # (the run with AGC_HIP_PACK_LOOKBACK=1 recorded as c_bench_lookback.json used the one-pass variant removed after commit 9ace5d7)
  timeout 900 python bench.py --config c5slice --c5-samples ${C5N:-128} > $OUT/c_bench_config_c5slice.json 2> $OUT/c_bench_config_c5slice.err; show $OUT/c_bench_config_c5slice.json
  timeout 300 python bench.py --config c5twin > $OUT/c_bench_config_c5twin.json 2> /dev/null; show $OUT/c_bench_config_c5twin.json
fi
if [ "$PART" = d ]; then
  timeout 300 python -m pytest tests/test_gpu_lz.py -m gpu -x -q -k "not every_launch" > $OUT/d_lz_tests.log 2>&1; tail -15 $OUT/d_lz_tests.log
  timeout 1200 python -m pytest tests/test_gpu_lz.py -m gpu -x -q -k "every_launch" > $OUT/d_lz_forced_chunks.log 2>&1; tail -25 $OUT/d_lz_forced_chunks.log
  timeout 200 python scripts/pack_alone.py 3.0 0 > $OUT/d_pack_alone.log 2>&1; tail -6 $OUT/d_pack_alone.log
  tail -6 $OUT/d_pack_alone.log
  timeout 300 python scripts/pack_alone.py 3.0 120 >> $OUT/d_pack_alone.log 2>&1; tail -6 $OUT/d_pack_alone.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/d_bench_driver_cmd_no_cpu_baseline.json 2> $OUT/d_bench_driver_cmd.err; show $OUT/d_bench_driver_cmd_no_cpu_baseline.json
  python - $OUT/d_bench_driver_cmd_no_cpu_baseline.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    v = d["config"].get("pack_ms_cumulative_after_each_pack") or []
    print("pack ms each:", [round(b - a, 2) for a, b in zip([0] + v[:-1], v)])
except Exception as e:
    print("no line", e)
PY
  timeout 900 python bench.py --config c5slice --c5-samples ${C5N:-128} > $OUT/d_bench_config_c5slice.json 2> $OUT/d_bench_config_c5slice.err; show $OUT/d_bench_config_c5slice.json
fi
if [ "$PART" = e ]; then
  timeout 300 python -m pytest tests/test_gpu_lz.py -m gpu -x -q -k "not every_launch" > $OUT/e_lz_tests.log 2>&1; tail -5 $OUT/e_lz_tests.log
  timeout 600 python -m pytest tests/test_gpu_lz.py -m gpu -x -q -k "every_launch and 300" > $OUT/e_lz_forced_chunks.log 2>&1; tail -5 $OUT/e_lz_forced_chunks.log
  bash scripts/c5_chunk_log.sh > $OUT/e_c5_chunk_log_summary.txt 2>&1; grep "launch mode" $OUT/e_c5_chunk_log.txt | awk '{print $3, $NF, $(NF-1), $(NF-5), $(NF-4)}' | sort | uniq -c | sort -k1nr | head -5; grep "launch mode [12]" $OUT/e_c5_chunk_log.txt | sed 's/.*chunk kernel/chunk kernel/' | sort -k7n | tail -5
  timeout 900 python bench.py --config c5slice --c5-samples ${C5N:-128} > $OUT/e_bench_config_c5slice.json 2> $OUT/e_bench_config_c5slice.err; show $OUT/e_bench_config_c5slice.json
  timeout 300 python bench.py --config c5twin > $OUT/e_bench_config_c5twin.json 2> /dev/null; show $OUT/e_bench_config_c5twin.json
  timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/e_bench_steps10.json 2> /dev/null; show $OUT/e_bench_steps10.json
fi
if [ "$PART" = f ]; then
  timeout 300 python -m pytest tests/test_gpu_scan.py tests/test_gpu_archive.py -m gpu -x -q -k "pack_fasta or small_samples_too" > $OUT/f_tests.log 2>&1; tail -4 $OUT/f_tests.log
  timeout 200 python scripts/pack_alone.py 3.0 0 > $OUT/f_pack_alone.log 2>&1; tail -3 $OUT/f_pack_alone.log
  timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/f_bench_steps10.json 2> /dev/null; show $OUT/f_bench_steps10.json
  timeout 900 python bench.py --config c5slice --c5-samples ${C5N:-128} > $OUT/f_bench_config_c5slice.json 2> $OUT/f_bench_config_c5slice.err; show $OUT/f_bench_config_c5slice.json
  for cfg in c5twin c1; do timeout 300 python bench.py --config $cfg > $OUT/f_bench_config_$cfg.json 2> /dev/null; show $OUT/f_bench_config_$cfg.json; done
fi
if [ "$PART" = g ]; then
  # ---- end of round, part 1: tests, smoke, the driver's command (with the CPU baseline), controls, configs, two ranks on one GPU, fuzz
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd_steps20_warmup5.json 2> $OUT/bench_driver_cmd.err; show $OUT/bench_driver_cmd_steps20_warmup5.json
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd_steps20_warmup5_run2.json 2>/dev/null; show $OUT/bench_driver_cmd_steps20_warmup5_run2.json
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prepacked > $OUT/bench_prepacked_control_steps20_warmup5.json 2>/dev/null; show $OUT/bench_prepacked_control_steps20_warmup5.json
  for cfg in c1 c4twin c5twin; do timeout 400 python bench.py --config $cfg > $OUT/bench_config_$cfg.json 2> /dev/null; show $OUT/bench_config_$cfg.json; done
  timeout 900 python bench.py --config c5slice --c5-samples 192 > $OUT/bench_config_c5slice.json 2> /dev/null; show $OUT/bench_config_c5slice.json
  AGC_BENCH_ONE_GPU=1 AGC_BENCH_SERIAL_PREPARE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks_serial_prepare.json 2> $OUT/bench_one_gpu_2_ranks_serial_prepare.err; show $OUT/bench_one_gpu_2_ranks_serial_prepare.json
  AGC_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks.json 2> $OUT/bench_one_gpu_2_ranks.err; show $OUT/bench_one_gpu_2_ranks.json
  # a Close of 6 300 full packs of 44 KB: what one of 8 GPUs holds when 200 human-size samples close: 104 samples of 378 Mbp, b = 100
  timeout 600 python bench.py --gbp 0.378 --steps 100 --warmup 4 --prepacked --no-cpu-baseline > $OUT/bench_6300_groups_104_samples_full_packs.json 2>/dev/null; show $OUT/bench_6300_groups_104_samples_full_packs.json
  timeout 900 python scripts/fuzz_archives.py --from 70000 --count 120 > $OUT/fuzz_gpu_120_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_120_cases.log
  timeout 600 python scripts/fuzz_archives.py --many --from 71000 --count 30 > $OUT/fuzz_gpu_many_30_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_many_30_cases.log
  timeout 600 python scripts/fuzz_archives.py --big --from 72000 --count 20 > $OUT/fuzz_gpu_big_20_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_big_20_cases.log
fi
if [ "$PART" = h ]; then
  # ---- end of round, part 2: rocprofv3 passes of the driver's command and of the c5slice CLI run, laps, --verify-entropy
  bash scripts/profile_round.sh r5 > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log
  python - <<'PY'
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
from agc_amd import synth
td = "/dev/shm/c5prof"; os.makedirs(td, exist_ok=True)
rng = np.random.default_rng(5)
anc = synth.random_seq(rng, 5_000_000)
plasmids = [synth.random_seq(rng, int(rng.integers(20_000, 90_000))) for _ in range(12)]
with open(td + "/files.txt", "w") as fl:
    for i in range(48):
        ctg, nm = [synth.mutate(rng, anc, 0.025)], [f"NZ_CP{i:06d}.1 strain {i} chromosome"]
        if i:
            for pi in rng.permutation(12)[: int(rng.integers(0, 3))]:
                ctg.append(synth.mutate(rng, plasmids[int(pi)], 0.025)); nm.append(f"NZ_CP{i:06d}p{int(pi)}.1 plasmid")
        fn = f"{td}/GCF_{i:09d}.fa"; synth.to_fasta(fn, ctg, nm); fl.write(fn + "\n")
PY
  C5="agc_amd/bin/agc_amd create -a -t 16 -o /dev/shm/c5prof/o.agc $(cat /dev/shm/c5prof/files.txt | tr '\n' ' ')"
  ROOT=$(pwd)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/c5_ktrace -o kt -- $ROOT/$C5 > $ROOT/$OUT/c5_ktrace.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv --kernel-include-regex agc -d $ROOT/$OUT/c5_pmc_fetch -o pf -- $ROOT/$C5 > $ROOT/$OUT/c5_pmc_fetch.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv --kernel-include-regex agc -d $ROOT/$OUT/c5_pmc_write -o pw -- $ROOT/$C5 > $ROOT/$OUT/c5_pmc_write.log 2>&1)
  python scripts/pmc_summary.py $OUT/pmc_summary_c5slice.csv $OUT/c5_pmc_fetch $OUT/c5_pmc_write > $OUT/pmc_summary_c5slice.log 2>&1
  find $OUT -name '*counter_collection.csv' -size +4M -delete; find $OUT -name '*kernel_trace.csv' -size +8M -delete
  # a Close that holds 6 300 nearly full packs (99 samples, b = 100): the device's launch on what ONE of 8 GPUs closes when 200 samples end
  timeout 600 python bench.py --gbp 0.378 --steps 95 --warmup 4 --prepacked --no-cpu-baseline > $OUT/bench_6300_groups_99_samples_close_holds_full_packs.json 2>/dev/null; show $OUT/bench_6300_groups_99_samples_close_holds_full_packs.json
  AGC_AMD_GPU_ZSTD_SHARE=1.0 timeout 600 python bench.py --gbp 0.378 --steps 95 --warmup 4 --prepacked --no-cpu-baseline > $OUT/bench_6300_groups_99_samples_all_packs_on_the_device.json 2>/dev/null; show $OUT/bench_6300_groups_99_samples_all_packs_on_the_device.json
  AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/bench_laps_steps20_warmup5.txt; grep -c lap $OUT/bench_laps_steps20_warmup5.txt
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --verify-entropy > $OUT/bench_verify_entropy.json 2> $OUT/bench_verify_entropy.log; grep -h "verify" $OUT/bench_verify_entropy.log | tail -3
  find $OUT -name "*kernel_stats.csv" | head; find $OUT -name "*kernel_stats.csv" -exec head -12 {} \;
fi
if [ "$PART" = i ]; then
  # ---- the last pass on the final build: tests, smoke, the driver's command, its control, the profiles, the configs
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd_steps20_warmup5.json 2> $OUT/bench_driver_cmd.err; show $OUT/bench_driver_cmd_steps20_warmup5.json
  for i in 2 3; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd_steps20_warmup5_run$i.json 2>/dev/null; show $OUT/bench_driver_cmd_steps20_warmup5_run$i.json; done
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prepacked > $OUT/bench_prepacked_control_steps20_warmup5.json 2>/dev/null; show $OUT/bench_prepacked_control_steps20_warmup5.json
  timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_default_no_cpu_baseline.json 2>/dev/null; show $OUT/bench_default_no_cpu_baseline.json
  timeout 200 python scripts/pack_alone.py 3.0 0 > $OUT/pack_alone.log 2>&1; tail -2 $OUT/pack_alone.log
  bash scripts/profile_round.sh r5 > $OUT/profile_round.log 2>&1
  for cfg in c1 c4twin c5twin; do timeout 400 python bench.py --config $cfg > $OUT/bench_config_$cfg.json 2> /dev/null; show $OUT/bench_config_$cfg.json; done
  timeout 900 python bench.py --config c5slice --c5-samples 192 > $OUT/bench_config_c5slice.json 2> /dev/null; show $OUT/bench_config_c5slice.json
  AGC_BENCH_ONE_GPU=1 AGC_BENCH_SERIAL_PREPARE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks_serial_prepare.json 2> $OUT/bench_one_gpu_2_ranks_serial_prepare.err; show $OUT/bench_one_gpu_2_ranks_serial_prepare.json
  find $OUT -name "*kernel_stats.csv" | head -3
fi
ls $OUT | head -100
if [ "$PART" = z ]; then
  # the last build of the round (experiment variants removed): tests, smoke, pack alone, the profile passes, the driver's bench line
  timeout 700 python -m pytest tests -m gpu -x -q -k "not every_launch" > $OUT/z_gpu_tests.log 2>&1; tail -4 $OUT/z_gpu_tests.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/z_smoke.log 2>&1; tail -2 $OUT/z_smoke.log
  timeout 200 python scripts/pack_alone.py 3.0 0 > $OUT/z_pack_alone.log 2>&1; tail -4 $OUT/z_pack_alone.log
  bash scripts/profile_round.sh r5/z_prof > $OUT/z_profile_round.log 2>&1; tail -3 $OUT/z_profile_round.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/z_bench_driver_cmd.json 2> $OUT/z_bench_driver_cmd.err; show $OUT/z_bench_driver_cmd.json
fi
if [ "$PART" = y ]; then
  # the host entropy stage without a barrier between batches (AGC_AMD_ENTROPY_STREAM), configs[1]; where the fixed start goes
  timeout 120 python scripts/start_cost.py > $OUT/y_start_cost.log 2>&1; cat $OUT/y_start_cost.log
  AGC_AMD_ENTROPY_STREAM=0 timeout 300 python bench.py --config c1 > $OUT/y_bench_config_c1_batches.json 2> /dev/null; show $OUT/y_bench_config_c1_batches.json
  timeout 300 python bench.py --config c1 > $OUT/y_bench_config_c1.json 2> /dev/null; show $OUT/y_bench_config_c1.json
  timeout 300 python bench.py --config c4twin > $OUT/y_bench_config_c4twin.json 2> /dev/null; show $OUT/y_bench_config_c4twin.json
fi
if [ "$PART" = x ]; then
  # the config lines again: host entropy stream, small read buffers, grouped small files
  for c in c1 c4twin c5twin; do timeout 300 python bench.py --config $c > $OUT/x_bench_config_$c.json 2> /dev/null; show $OUT/x_bench_config_$c.json; done
  timeout 900 python bench.py --config c5slice > $OUT/x_bench_config_c5slice.json 2> $OUT/x_bench_config_c5slice.err; show $OUT/x_bench_config_c5slice.json
fi
if [ "$PART" = w ]; then
  # the round's last host code: every GPU test, smoke, fuzz through the real kernels, the driver's bench line
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/w_gpu_tests.log 2>&1; tail -n 3 $OUT/w_gpu_tests.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/w_smoke.log 2>&1; tail -n 2 $OUT/w_smoke.log
  timeout 400 python scripts/fuzz_archives.py --from 9500 --count 60 > $OUT/w_fuzz_gpu_60_cases.log 2>&1; tail -n 2 $OUT/w_fuzz_gpu_60_cases.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/w_bench_driver_cmd.json 2> $OUT/w_bench_driver_cmd.err; show $OUT/w_bench_driver_cmd.json
fi
if [ "$PART" = v ]; then
  # last evidence on the round's last code: every device frame of the bench's Close against libzstd; fuzz flavours through the real kernels
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --verify-entropy > $OUT/v_bench_verify_entropy.json 2> $OUT/v_bench_verify_entropy.log; grep -h "verify" $OUT/v_bench_verify_entropy.log | tail -n 3
  timeout 240 python scripts/fuzz_archives.py --many --from 9700 --count 20 > $OUT/v_fuzz_gpu_many_20_cases.log 2>&1; tail -n 1 $OUT/v_fuzz_gpu_many_20_cases.log
  timeout 240 python scripts/fuzz_archives.py --big --from 9800 --count 15 > $OUT/v_fuzz_gpu_big_15_cases.log 2>&1; tail -n 1 $OUT/v_fuzz_gpu_big_15_cases.log
fi
if [ "$PART" = u ]; then
  # the FASTA conversion's stream at the lowest priority (AGC_HIP_PACK_LOW_PRIORITY existed for this call only: profiles/EXPERIMENTS.md; not kept)
  for i in 1 2; do
    AGC_HIP_PACK_LOW_PRIORITY=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/u_bench_pack_low_priority_$i.json 2> /dev/null; show $OUT/u_bench_pack_low_priority_$i.json
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/u_bench_pack_default_priority_$i.json 2> /dev/null; show $OUT/u_bench_pack_default_priority_$i.json
  done
fi
if [ "$PART" = t ]; then
  # kernel timeline of the driver's command: start / end of every kernel, for the busy / idle accounting of a step (scripts/step_timeline.py)
  export TMPDIR=/tmp; ROOT=$(pwd)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/t_ktrace -o kt -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/$OUT/t_ktrace.log 2>&1)
  tail -n 1 $OUT/t_ktrace.log | cut -c1-200
  find $OUT/t_ktrace -name "*kernel_trace.csv" -exec ls -la {} \;
fi
if [ "$PART" = s ]; then
  # the conversion as a few blocks per CU walking the tiles (AGC_HIP_PACK_BLOCKS_PER_CU existed for this call only: profiles/EXPERIMENTS.md; reverted)
  timeout 200 python -m pytest tests/test_gpu_scan.py -m gpu -x -q -k "pack_fasta" > $OUT/s_pack_tests.log 2>&1; tail -n 2 $OUT/s_pack_tests.log
  for n in 2 1000; do AGC_HIP_PACK_BLOCKS_PER_CU=$n timeout 100 python scripts/pack_alone.py 3.0 0 > $OUT/s_pack_alone_$n.log 2>&1; echo "per CU $n:"; tail -n 2 $OUT/s_pack_alone_$n.log; done
  for i in 1 2; do for n in 2 1000 1 4; do
    AGC_HIP_PACK_BLOCKS_PER_CU=$n timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/s_bench_pack_${n}_per_cu_$i.json 2> /dev/null; show $OUT/s_bench_pack_${n}_per_cu_$i.json
  done; done
fi
