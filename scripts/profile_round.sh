#!/bin/bash
# scripts/profile_round.sh TAG -- the rocprofv3 evidence of one round, written under gpurun_out/TAG/ on the GPU box
# (copy what should be judged into profiles/TAG/).  Separate passes, as MI355X_MICROARCH.md prescribes:
#   1. --kernel-trace --stats of bench.py (falls back to scripts/quick_bench.py if the profiler dies on bench.py)
#   2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE   (bench.py --steps 2 --warmup 1, kernels of namespace agc only)
#   4. FETCH_SIZE calibration micro-kernels (scripts/fetch_calib)
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
B="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
# the PMC passes run the driver's own shape (25 samples: the packs of Close() are what the entropy kernel really sees)
BP="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/ktrace -o kt -- $BP > $ROOT/$OUT/ktrace.log 2>&1)
if [ -z "$(find $OUT/ktrace -name "*kernel_stats.csv" 2>/dev/null)" ]; then
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/ktrace_qb -o kt -- python $ROOT/scripts/quick_bench.py 3e9 > $ROOT/$OUT/ktrace_qb.log 2>&1)
fi
(cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv --kernel-include-regex agc -d $ROOT/$OUT/pmc_fetch -o pf -- $BP > $ROOT/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv --kernel-include-regex agc -d $ROOT/$OUT/pmc_write -o pw -- $BP > $ROOT/$OUT/pmc_write.log 2>&1)
python scripts/pmc_summary.py $OUT/pmc_summary.csv $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.log 2>&1
if [ -x scripts/fetch_calib ]; then
  (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv --kernel-include-regex calib -d $ROOT/$OUT/calib -o cal -- $ROOT/scripts/fetch_calib > $ROOT/$OUT/fetch_calib.txt 2>&1)
  python scripts/pmc_summary.py $OUT/fetch_calib_pmc.csv $OUT/calib >> $OUT/fetch_calib.txt 2>&1
  cat $OUT/fetch_calib_pmc.csv >> $OUT/fetch_calib.txt
fi
python scripts/step_timeline.py $(find $OUT/ktrace -name "*kernel_trace.csv" | head -1) > $OUT/step_timeline.txt 2>&1
# keep the merged-back volume small: the raw per-dispatch CSVs of the PMC passes are large
find $OUT -name '*counter_collection.csv' -size +4M -delete
find $OUT -name '*kernel_trace.csv' -size +8M -delete
ls -la $OUT
