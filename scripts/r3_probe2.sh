#!/bin/bash
# round 3: zstd group kernel variants (tiny parity first, short timeouts) + stall counters of the default
OUT=gpurun_out/r3p2
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
timeout 120 python scripts/r3_tiny.py 3 2 0 > $OUT/tiny.log 2>&1 || { echo "TINY FAILED"; tail -5 $OUT/tiny.log; exit 1; }
grep group $OUT/tiny.log
timeout 300 python -m pytest tests/test_gpu_zstd.py -x -q > $OUT/test_gpu_zstd.log 2>&1
tail -2 $OUT/test_gpu_zstd.log
for G in 3 2; do
  AGC_HIP_ZSTD_GROUP=$G timeout 120 python scripts/zstd_gpu_probe.py 36000 real > $OUT/probe_g$G.log 2>&1
  echo "G=$G W=12: $(grep 'run 1' $OUT/probe_g$G.log) $(grep -c identical $OUT/probe_g$G.log)"
done
for W in 8 16; do
  AGC_HIP_LIB=$ROOT/scripts/variants/libagc_hip_gw$W.so AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py 36000 real > $OUT/probe_g3_w$W.log 2>&1
  echo "G=3 W=$W: $(grep 'run 1' $OUT/probe_g3_w$W.log) $(grep -c identical $OUT/probe_g3_w$W.log)"
done
AGC_HIP_ZSTD_GROUP=3 timeout 120 python scripts/zstd_gpu_probe.py 42000 real > $OUT/probe_g3_42k.log 2>&1
echo "G=3 42k frames: $(grep 'run 1' $OUT/probe_g3_42k.log)"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && AGC_HIP_ZSTD_GROUP=3 timeout 150 rocprofv3 --pmc $set --output-format csv --kernel-include-regex zstd_frames -d $ROOT/$OUT/pmc$i -o p -- python $ROOT/scripts/zstd_gpu_probe.py 36000 real > $ROOT/$OUT/pmc$i.log 2>&1)
done
python scripts/pmc_summary.py $OUT/pmc_summary.csv $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 > /dev/null 2>&1
cat $OUT/pmc_summary.csv
find $OUT -name '*counter_collection.csv' -size +2M -delete
