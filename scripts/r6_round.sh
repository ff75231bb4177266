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
if [ "$PART" = d ]; then
  # stream priorities (0 off, 1 steps high + bulk low, 2 steps high only); host threads of the entropy pool; the new thread pool
  bench d_bench_prio1
  bench d_bench_prio0 AGC_HIP_STREAM_PRIORITIES=0
  bench d_bench_prio2 AGC_HIP_STREAM_PRIORITIES=2
  bench d_bench_prio1_b
  bench d_bench_prio0_b AGC_HIP_STREAM_PRIORITIES=0
  for t in 20 24; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --threads $t > $OUT/d_bench_threads$t.json 2>/dev/null; show $OUT/d_bench_threads$t.json; done
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/d_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["config"]["step_ms_each_rank0"], d["config"].get("cgroup_cpu"))
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = e ]; then
  # Close: the pool starts at once, the device's side on its own thread; arena chunks grow geometrically; no stream priorities
  timeout 600 python -m pytest tests/test_gpu_archive.py tests/test_gpu_zstd.py -m gpu -x -q > $OUT/e_tests.log 2>&1; tail -3 $OUT/e_tests.log
  bench e_bench_1
  bench e_bench_2
  bench e_bench_3
  AGC_AMD_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/e_bench_laps.json 2> $OUT/e_bench_laps.txt; show $OUT/e_bench_laps.json
  grep -n "close lap\|entropy lap\|finish_groups" $OUT/e_bench_laps.txt | tail -12
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/e_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["config"]["step_ms_each_rank0"], d["config"].get("cgroup_cpu"))
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = f ]; then
  # bulk streams on a CU mask (32 CUs never take a bulk block); S1a on the file path; allocation latencies
  python scripts/malloc_probe.py > $OUT/f_malloc_probe.txt 2>&1; cat $OUT/f_malloc_probe.txt
  timeout 900 python -m pytest tests/test_gpu_archive.py -m gpu -x -q > $OUT/f_archive_tests.log 2>&1; tail -3 $OUT/f_archive_tests.log
  bench f_bench_mask32
  bench f_bench_mask0 AGC_HIP_BULK_FREE_CUS=0
  bench f_bench_mask64 AGC_HIP_BULK_FREE_CUS=64
  bench f_bench_mask32_b
  bench f_bench_mask0_b AGC_HIP_BULK_FREE_CUS=0
  AGC_AMD_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/f_bench_laps.json 2> $OUT/f_bench_laps.txt; show $OUT/f_bench_laps.json
  timeout 600 python bench.py --from-fasta 2 > $OUT/f_from_fasta.json 2> $OUT/f_from_fasta.err; cat $OUT/f_from_fasta.json | cut -c1-700
  AGC_AMD_FASTA_PACK=0 timeout 600 python bench.py --from-fasta 2 > $OUT/f_from_fasta_off.json 2> $OUT/f_from_fasta_off.err; cat $OUT/f_from_fasta_off.json | cut -c1-700
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/f_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["config"]["step_ms_each_rank0"], d["config"].get("cgroup_cpu"))
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = g ]; then
  # pack kernel with byte-permute look-ups and bit-matrix transposes; exact escape flags; Close with the gather by the pool again
  timeout 600 python -m pytest tests/test_gpu_scan.py -m gpu -x -q -k "pack_fasta" > $OUT/g_pack_tests.log 2>&1; tail -5 $OUT/g_pack_tests.log
  timeout 200 python scripts/pack_alone.py 3.0 0 > $OUT/g_pack_alone.log 2>&1; tail -4 $OUT/g_pack_alone.log
  timeout 900 python -m pytest tests/test_gpu_archive.py -m gpu -x -q > $OUT/g_archive_tests.log 2>&1; tail -3 $OUT/g_archive_tests.log
  bench g_bench_1
  bench g_bench_2
  bench g_bench_3
  AGC_AMD_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/g_bench_laps.json 2> $OUT/g_bench_laps.txt; show $OUT/g_bench_laps.json
  grep -n "close lap\|entropy lap\|finish_groups" $OUT/g_bench_laps.txt | tail -12
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/g_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["config"]["step_ms_each_rank0"], d["config"].get("cgroup_cpu"))
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = h ]; then
  # where the hiccups of the FIRST bench run on a fresh box come from (steps 9 and 17 of the timed region: 180 + 42 ms): laps first
  AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/h_bench_laps_first.json 2> $OUT/h_bench_laps_first.txt; show $OUT/h_bench_laps_first.json
  python - <<'PY'
import re, collections, json
lines = open("gpurun_out/r6/h_bench_laps_first.txt").read().splitlines()
steps = []; cur = None
for l in lines:
    m = re.match(r'\s+lap (.*?) ([0-9.e+-]+) ms$', l)
    if m:
        if m.group(1) == 'group map -> device':
            cur = collections.OrderedDict(); steps.append(cur)
        if cur is not None:
            cur[m.group(1)] = cur.get(m.group(1), 0) + float(m.group(2))
for i, s in enumerate(steps[-20:]):
    tot = sum(s.values())
    if tot > 17:
        print("timed step", i, "sum", round(tot, 1), {k: round(v, 1) for k, v in s.items() if v > 1.5})
d = json.loads(open("gpurun_out/r6/h_bench_laps_first.json").read().strip().splitlines()[-1])
print(d["config"]["step_ms_each_rank0"])
PY
  grep -n "prepare_batch lap\|lz_encode_end lap\|segments_packed lap" $OUT/h_bench_laps_first.txt | awk '{ if ($(NF-1)+0 > 8) print }' | head -20
  bench h_bench_2
fi
if [ "$PART" = i ]; then
  AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/i_bench_laps_first.json 2> $OUT/i_bench_laps_first.txt; show $OUT/i_bench_laps_first.json
  grep -n "ensure:\|arena:" $OUT/i_bench_laps_first.txt | tail -40
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/i_bench_laps_first.json").read().strip().splitlines()[-1])
print(d["config"]["step_ms_each_rank0"]); print(d["config"]["setup_not_timed"])
PY
fi
if [ "$PART" = j ]; then
  # the arena's next chunk allocated ahead by a helper thread: the first run on a fresh box again
  AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/j_bench_first.json 2> $OUT/j_bench_first.txt; show $OUT/j_bench_first.json
  grep -n "ensure:\|arena:" $OUT/j_bench_first.txt | tail -12
  bench j_bench_2
  bench j_bench_3
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/j_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["config"]["step_ms_each_rank0"], d["config"].get("cgroup_cpu"))
    except Exception as e:
        print(n, e)
PY
  timeout 900 python -m pytest tests/test_dist_single_archive.py -m gpu -x -q -k "rccl" > $OUT/j_rccl_one_rank.log 2>&1; tail -5 $OUT/j_rccl_one_rank.log
fi
if [ "$PART" = k ]; then
  # lag counters + symbols of the new references in one submission nobody on the steps' thread waits for
  timeout 900 python -m pytest tests/test_gpu_archive.py -m gpu -x -q > $OUT/k_archive_tests.log 2>&1; tail -3 $OUT/k_archive_tests.log
  bench k_bench_1
  bench k_bench_off AGC_AMD_REF_STORE_ASYNC=0
  bench k_bench_2
  bench k_bench_3
  AGC_AMD_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/k_bench_laps.json 2> $OUT/k_bench_laps.txt; show $OUT/k_bench_laps.json
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/k_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["config"]["step_ms_each_rank0"])
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = l ]; then
  # the same with the context's packed buffers protected from the next window's pack; where the start of a run goes
  timeout 900 python -m pytest tests/test_gpu_archive.py -m gpu -x -q > $OUT/l_archive_tests.log 2>&1; tail -3 $OUT/l_archive_tests.log
  timeout 900 python -m pytest tests/test_gpu_archive.py -m gpu -x -q > $OUT/l_archive_tests2.log 2>&1; tail -3 $OUT/l_archive_tests2.log
  python - <<'PY'
import os, subprocess, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.getcwd())
from agc_amd import synth
td = tempfile.mkdtemp(dir="/dev/shm")
fn = td + "/tiny.fa"
synth.to_fasta(fn, [synth.random_seq(np.random.default_rng(1), 2000)], ["tiny"])
for rep in range(3):
    t0 = time.perf_counter()
    r = subprocess.run(["agc_amd/bin/agc_amd", "create", "-o", td + "/t.agc", fn], capture_output=True, text=True, env=dict(os.environ, AGC_AMD_START_LAPS="1"))
    print(f"CLI wall {time.perf_counter() - t0:.3f} s"); print(r.stderr)
PY
  bench l_bench_1
  bench l_bench_2
  for cfg in c1; do timeout 400 python bench.py --config $cfg > $OUT/l_bench_config_$cfg.json 2> /dev/null; show $OUT/l_bench_config_$cfg.json; done
fi
if [ "$PART" = m ]; then
  timeout 600 python scripts/c1_probe.py > $OUT/m_c1_probe.txt 2>&1; cat $OUT/m_c1_probe.txt
fi
if [ "$PART" = n ]; then
  # deals in the middle of an N-rank run with the real kernels; two ranks on one GPU with packs that fill during the steps
  timeout 900 python -m pytest tests/test_dist_single_archive.py -m gpu -x -q > $OUT/n_dist_gpu_tests.log 2>&1; tail -3 $OUT/n_dist_gpu_tests.log
  for dm in -1 0; do
    AGC_AMD_DEAL_MIN_MB=$dm AGC_AMD_DEAL_EVERY=2 AGC_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2961$((dm+2)) bench.py --gpus 2 --steps 12 --warmup 2 --pack-cardinality 8 --no-cpu-baseline > $OUT/n_bench_2ranks_b8_deal$dm.json 2> $OUT/n_bench_2ranks_b8_deal$dm.err; show $OUT/n_bench_2ranks_b8_deal$dm.json
    python - $OUT/n_bench_2ranks_b8_deal$dm.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  ms per sample rank0:", d["config"].get("single_archive_ms_per_sample_rank0")); print("  ", d["config"].get("parallelism")[-420:]); print("  zstd", d["config"]["zstd"])
except Exception as e:
    print("no line", e)
PY
    tail -3 $OUT/n_bench_2ranks_b8_deal$dm.err
  done
fi
if [ "$PART" = o ]; then
  # the encode kernel with room for 7 / 8 waves per SIMD (71 / 64 VGPRs; default 78 -> 6 waves)
  timeout 300 python -m pytest tests/test_gpu_lz.py -m gpu -x -q -k "not every_launch" > $OUT/o_lz_tests_base.log 2>&1; tail -2 $OUT/o_lz_tests_base.log
  for v in lz_w8 lz_w7; do
    LD_PRELOAD=$PWD/scripts/variants/$v.so AGC_HIP_LIB=$PWD/scripts/variants/$v.so timeout 300 python -m pytest tests/test_gpu_lz.py -m gpu -x -q -k "not every_launch" > $OUT/o_lz_tests_$v.log 2>&1; tail -2 $OUT/o_lz_tests_$v.log
  done
  for rep in 1 2; do
    bench o_bench_base_$rep
    for v in lz_w8 lz_w7; do bench o_bench_${v}_$rep LD_PRELOAD=$PWD/scripts/variants/$v.so AGC_HIP_LIB=$PWD/scripts/variants/$v.so; done
  done
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/o_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        e = d["config"]["step_ms_each_rank0"]
        print(n.split("/")[-1], "median step", sorted(e)[len(e) // 2], e)
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = p ]; then
  # where a registration queues its whole-sample encode: 0 at once (beside estimates + cost vectors), 1 behind the estimates, 2 behind the classification (beside scan + conversion)
  for m in 1 2; do AGC_AMD_ENCODE_LAUNCH_AT=$m timeout 600 python -m pytest tests/test_gpu_archive.py -m gpu -x -q -k "c3_twin or small_samples or full_size" > $OUT/p_tests_at$m.log 2>&1; tail -1 $OUT/p_tests_at$m.log; done
  for rep in 1 2; do for m in 0 1 2; do bench p_bench_at${m}_$rep AGC_AMD_ENCODE_LAUNCH_AT=$m; done; done
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/p_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        e = d["config"]["step_ms_each_rank0"]
        print(n.split("/")[-1], "median step", sorted(e)[len(e) // 2], e)
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = z1 ]; then
  # ---- end of round, part 1: tests, smoke, the driver's command (with the CPU baseline), controls, configs, two ranks on one GPU, fuzz
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd_steps20_warmup5.json 2> $OUT/bench_driver_cmd.err; show $OUT/bench_driver_cmd_steps20_warmup5.json
  for i in 2 3; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd_steps20_warmup5_run$i.json 2>/dev/null; show $OUT/bench_driver_cmd_steps20_warmup5_run$i.json; done
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prepacked > $OUT/bench_prepacked_control_steps20_warmup5.json 2>/dev/null; show $OUT/bench_prepacked_control_steps20_warmup5.json
  timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_default_no_cpu_baseline.json 2>/dev/null; show $OUT/bench_default_no_cpu_baseline.json
  for cfg in c1 c4twin c5twin; do timeout 400 python bench.py --config $cfg > $OUT/bench_config_$cfg.json 2> /dev/null; show $OUT/bench_config_$cfg.json; done
  timeout 900 python bench.py --config c5slice --c5-samples 192 > $OUT/bench_config_c5slice.json 2> /dev/null; show $OUT/bench_config_c5slice.json
  AGC_BENCH_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks.json 2> $OUT/bench_one_gpu_2_ranks.err; show $OUT/bench_one_gpu_2_ranks.json
  timeout 900 python scripts/fuzz_archives.py --from 90000 --count 120 > $OUT/fuzz_gpu_120_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_120_cases.log
  timeout 600 python scripts/fuzz_archives.py --many --from 91000 --count 30 > $OUT/fuzz_gpu_many_30_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_many_30_cases.log
  timeout 600 python scripts/fuzz_archives.py --big --from 92000 --count 20 > $OUT/fuzz_gpu_big_20_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_big_20_cases.log
  timeout 900 python scripts/fuzz_deals_gpu.py --from 93000 --count 24 > $OUT/fuzz_gpu_deals_24_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_deals_24_cases.log
fi
if [ "$PART" = z2 ]; then
  # ---- end of round, part 2: rocprofv3 passes of the driver's command and of the c5slice CLI run, laps, --verify-entropy, pack counters, deals fuzz again
  bash scripts/profile_round.sh r6 > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log; cat $OUT/step_timeline.txt | head -12
  timeout 200 python scripts/pack_alone.py 3.0 0 > $OUT/pack_alone.log 2>&1; tail -3 $OUT/pack_alone.log
  bash scripts/pack_pmc_probe.sh > $OUT/pack_pmc_probe.log 2>&1; tail -5 $OUT/pack_pmc_probe.log
  python - <<'PY'
import os, sys, numpy as np
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
  AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_laps.json 2> $OUT/bench_laps_steps20_warmup5.txt; show $OUT/bench_laps.json
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --verify-entropy > $OUT/bench_verify_entropy.json 2> $OUT/bench_verify_entropy.log; grep -h "verify" $OUT/bench_verify_entropy.log | tail -3
  timeout 900 python scripts/fuzz_deals_gpu.py --from 93000 --count 30 > $OUT/fuzz_gpu_deals_30_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_deals_30_cases.log
  AGC_BENCH_ONE_GPU=1 AGC_AMD_DEAL_MIN_MB=-1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_one_gpu_2_ranks_no_deals.json 2> /dev/null; show $OUT/bench_one_gpu_2_ranks_no_deals.json
  find $OUT -name "*kernel_stats.csv" | head -3
fi
if [ "$PART" = q ]; then
  # the chunk logs allocated ahead: the first run on a fresh box once more (laps: which buffers grow inside steps), then the GPU suite on the final build
  AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/q_bench_first.json 2> $OUT/q_bench_first.txt; show $OUT/q_bench_first.json
  grep -n "ensure:\|arena:" $OUT/q_bench_first.txt | tail -8
  bench q_bench_2
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/q_gpu_tests.log 2>&1; tail -3 $OUT/q_gpu_tests.log
  bench q_bench_3
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/q_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["config"]["step_ms_each_rank0"])
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = r ]; then
  AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r_bench_first.json 2> $OUT/r_bench_first.txt; show $OUT/r_bench_first.json
  grep -n "ensure:\|arena:" $OUT/r_bench_first.txt | tail -5
  bench r_bench_2
  timeout 600 python -m pytest tests/test_gpu_lz.py tests/test_gpu_archive.py -m gpu -x -q -k "not every_launch" > $OUT/r_tests.log 2>&1; tail -2 $OUT/r_tests.log
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/r_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["config"]["step_ms_each_rank0"])
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = s ]; then
  # timers read later instead of a wait behind every kernel; no wait behind the index build
  timeout 1500 python -m pytest tests -m gpu -q > $OUT/s_tests.log 2>&1; tail -2 $OUT/s_tests.log
  bench s_bench_1; bench s_bench_2; bench s_bench_3
  timeout 300 python bench.py --config c5twin > $OUT/s_c5twin.json 2>/dev/null; show $OUT/s_c5twin.json
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/s_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        e = d["config"]["step_ms_each_rank0"]
        print(n.split("/")[-1], "median", sorted(e)[len(e)//2], e)
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = t ]; then
  # what a step that takes twice as long was waiting for: laps of the driving thread, three runs
  for i in 1 2 3; do
    AGC_AMD_LAPS=1 AGC_HIP_LAPS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/t_laps_$i.json 2> $OUT/t_laps_$i.txt; show $OUT/t_laps_$i.json
    python scripts/lap_outliers.py $OUT/t_laps_$i.txt | cut -c1-600
    python - $OUT/t_laps_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["config"]["step_ms_each_rank0"])
PY
  done
fi
if [ "$PART" = u ]; then
  timeout 1500 python -m pytest tests/test_gpu_lz.py tests/test_gpu_archive.py tests/test_gpu_scan.py -m gpu -x -q > $OUT/u_tests.log 2>&1; tail -2 $OUT/u_tests.log
fi
if [ "$PART" = v ]; then
  # what a large allocation on a helper thread costs the thread that drives kernels
  hipcc --offload-arch=gfx950 -O2 -w -o /tmp/msp scripts/malloc_stall_probe.hip -lpthread || exit 1
  { timeout 60 /tmp/msp 4 6; timeout 60 /tmp/msp 0 6; timeout 60 /tmp/msp 0 6; timeout 60 /tmp/msp 0 6 40; timeout 60 /tmp/msp 1 6 60; timeout 60 /tmp/msp 2 6 80
    timeout 60 /tmp/msp 3 6 100; timeout 60 /tmp/msp 0 6 120; timeout 60 /tmp/msp 0 1 140; timeout 60 /tmp/msp 0 12 150; } > $OUT/v_malloc_stall.txt 2>&1
  cat $OUT/v_malloc_stall.txt
fi
if [ "$PART" = w ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $OUT/w_tests.log 2>&1; tail -2 $OUT/w_tests.log
  bench w_bench_1; bench w_bench_2; bench w_bench_3
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/w_bench_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        e = d["config"]["step_ms_each_rank0"]
        print(n.split("/")[-1], "median", sorted(e)[len(e)//2], e)
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = x ]; then
  # the scan's LDS filter with fewer instructions per position: parity, the kernel alone, the step
  timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_archive.py -m gpu -x -q > $OUT/x_tests.log 2>&1; tail -2 $OUT/x_tests.log
  timeout 200 python scripts/scan_alone.py 3.0 > $OUT/x_scan_alone.txt 2>&1; tail -3 $OUT/x_scan_alone.txt
  bench x_bench_1; bench x_bench_2
fi
if [ "$PART" = y ]; then
  # key filter: 32-bit filter words, second filter only for the keys that pass the first; threads per copy of the filter
  for t in 256 512 1024 256 512 1024; do bench y_bench_t${t}_$RANDOM AGC_HIP_FILTER_THREADS=$t; done
fi
if [ "$PART" = gantt ]; then
  ROOT=$(pwd)
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/gantt_ktrace -o kt -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/$OUT/gantt_ktrace.log 2>&1)
  F=$(find $OUT/gantt_ktrace -name "*kernel_trace.csv" | head -1)
  python scripts/step_timeline.py $F > $OUT/gantt_step_timeline.txt 2>&1; head -30 $OUT/gantt_step_timeline.txt
  python scripts/step_gantt.py $F 5 > $OUT/gantt_step.txt 2>&1; cat $OUT/gantt_step.txt
  python scripts/step_gantt.py $F 9 >> $OUT/gantt_step.txt 2>&1
  find $OUT/gantt_ktrace -name '*kernel_trace.csv' -size +8M -delete
fi
if [ "$PART" = fuzzlast ]; then
  # more random collections through the real kernels on the round's last build (new seeds)
  timeout 2400 python scripts/fuzz_archives.py --from ${FUZZ_FROM:-95000} --count 400 > $OUT/fuzz_gpu_last_build_400_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_last_build_400_cases.log
  timeout 1200 python scripts/fuzz_archives.py --many --from $(( ${FUZZ_FROM:-95000} + 1000 )) --count 60 > $OUT/fuzz_gpu_last_build_many_60_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_last_build_many_60_cases.log
  timeout 1200 python scripts/fuzz_archives.py --big --from $(( ${FUZZ_FROM:-95000} + 2000 )) --count 40 > $OUT/fuzz_gpu_last_build_big_40_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_last_build_big_40_cases.log
  timeout 1500 python scripts/fuzz_deals_gpu.py --from $(( ${FUZZ_FROM:-95000} + 3000 )) --count 40 > $OUT/fuzz_gpu_last_build_deals_40_cases.log 2>&1; tail -2 $OUT/fuzz_gpu_last_build_deals_40_cases.log
fi
if [ "$PART" = pre ]; then
  # the whole-sample encode launched inside agc_hip_segments_packed (before the segment table comes over) or behind it: alternating on one box
  [ -n "$PRE_TESTS" ] && { timeout 1500 python -m pytest tests/test_gpu_archive.py tests/test_gpu_scan.py tests/test_dist_single_archive.py -m gpu -x -q > $OUT/pre_tests.log 2>&1; tail -2 $OUT/pre_tests.log; }
  for i in 1 2 3 4 5 6 7 8; do bench pre_on_$i AGC_AMD_PRE_LAUNCH_ENCODE=1; bench pre_off_$i AGC_AMD_PRE_LAUNCH_ENCODE=0; done
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/pre_o*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        e = d["config"]["step_ms_each_rank0"]
        print(n.split("/")[-1], "value", d["value"], "median step", sorted(e)[len(e)//2], "mean", round(sum(e)/len(e), 2), "close", d["config"]["close_ms"])
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = ahead ]; then
  # placement of the segments without a split-point job beside the wait for the split points: parity, then on / off alternating on one box
  timeout 1500 python -m pytest tests/test_gpu_archive.py tests/test_dist_single_archive.py -m gpu -x -q > $OUT/ahead_tests.log 2>&1; tail -2 $OUT/ahead_tests.log
  AGC_AMD_PLACE_AHEAD=2 timeout 900 python scripts/fuzz_archives.py --from 120000 --count 150 > $OUT/ahead_fuzz_150_cases.log 2>&1; tail -1 $OUT/ahead_fuzz_150_cases.log
  for i in 1 2 3 4 5 6; do bench ahead_on_$i AGC_AMD_PLACE_AHEAD=1; bench ahead_off_$i AGC_AMD_PLACE_AHEAD=0; done > /dev/null
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/ahead_o*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        e = d["config"]["step_ms_each_rank0"]
        print(n.split("/")[-1], "value", d["value"], "median step", sorted(e)[len(e)//2], "mean", round(sum(e)/len(e), 2), "close", d["config"]["close_ms"])
    except Exception as e:
        print(n, e)
PY
fi
if [ "$PART" = specfill ]; then
  # the table of speculative deltas filled by a helper thread (off the chain from the group look-up to the estimates): parity, then on / off
  timeout 1500 python -m pytest tests/test_gpu_archive.py tests/test_dist_single_archive.py -m gpu -x -q > $OUT/specfill_tests.log 2>&1; tail -2 $OUT/specfill_tests.log
  AGC_AMD_SPEC_FILL_AHEAD=2 AGC_AMD_PLACE_AHEAD=2 timeout 900 python scripts/fuzz_archives.py --from 130000 --count 150 > $OUT/specfill_fuzz_150_cases.log 2>&1; tail -1 $OUT/specfill_fuzz_150_cases.log
  for i in 1 2 3 4 5 6; do bench specfill_on_$i AGC_AMD_SPEC_FILL_AHEAD=1; bench specfill_off_$i AGC_AMD_SPEC_FILL_AHEAD=0; done > /dev/null
  python - <<'PY'
import json, glob
for n in sorted(glob.glob("gpurun_out/r6/specfill_o*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        e = d["config"]["step_ms_each_rank0"]
        print(n.split("/")[-1], "value", d["value"], "median step", sorted(e)[len(e)//2], "mean", round(sum(e)/len(e), 2), "close", d["config"]["close_ms"])
    except Exception as e:
        print(n, e)
PY
fi
